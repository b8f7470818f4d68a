// EKF covariance algebra for a batch of independent filters, one workgroup per filter.
//
// Replaces the dense parts of odometry::EKF (reference: src/odometry/ekf.cpp; oracle:
// oracle/ekf_oracle.c). State m (n) and covariance P (n x n, f64, column-major as Eigen) live in
// HBM; the mean-side scalar bookkeeping (sample times, augment counters, rate limits) stays in the
// host adapter. Design for CDNA4 (numbers: scripts/f64_ubench.hip, scripts/ekf_microbench.py):
//   * f64 vector and f64 matrix peak are the same 128 flop/clk/CU on gfx950; the matrix cores are
//     used because one v_mfma_f64_16x16x4_f64 retires 1024 MACs for two operand loads (one f64 per
//     lane each) where a v_fma_f64 needs two operands per MAC, and because a lone wavefront issues a
//     dependent v_fma_f64 only every ~7 cycles;
//   * visual update / chi2 gate (ekf_update_kernel): S = H P H' + R is factored and both triangular
//     solves are done in ONE blocked Cholesky pass over the tall matrix T = [S ; v' ; (HP)'] in LDS
//     (16-column blocks: MFMA panel / trailing updates, the 16 x 16 diagonal factor and its inverse
//     in the registers of one wavefront), after which chi2 = ns z'z, m += Y'z, P -= Y'Y with
//     Y = L^-1 HP. K is never formed; the reference forms K = (S^-1 HP)' with a pivoted LDLT and
//     P -= K HP, algebraically identical (parity 1e-9 relative per call, measured ~1e-13). At the
//     reference's sizes (n = 160, <= 47 rows) H is staged in LDS too and every wavefront owns whole
//     16-column blocks of P: the tiles it streams from HBM as MFMA operands of H P are already in the
//     accumulator layout of P -= Y'Y and stay in registers until then, so P is read once per update;
//   * predict: thread 0 evaluates the mean and the 20 x 20 Jacobians while the other wavefronts
//     already hold the slabs of P10 / P01 they need; the off-diagonal blocks are MFMA items;
//   * pose augmentation: the Joseph form (I-KH) P (I-KH)' + K R K' is expanded around the 7-row
//     +-1 matrix visAugH into a rank-14 correction (4 MFMA k-steps per 16 x 16 tile) of the shifted
//     matrix A P A' + Q, which is a gather of P and never materialised; mirrored tile pairs meet
//     through an in-wave LDS transpose for the fused (P+P')/2, and the result goes to the second of
//     two ping-pong buffers: one read and one write of P per augmentation.
#include <functional>
#include <limits.h>
#include <math.h>

#include <utility>

#include <stdlib.h>

#include "ekf_device.hpp"

// The library is built with -ffp-contract=off for the tracker's bit-exact binary32 sequence; the
// EKF is judged against a relative tolerance, and a fused multiply-add is one rounding fewer and
// half the f64 instructions of its dependency chains.
#pragma clang fp contract(fast)

namespace hv {

namespace {


// developer aid: s_memtime stamps of the kernels' phases (block 0, thread 0), read back through
// hv_debug_ekf_phase_stamps. Compiled in only with -DHV_EKF_PHASE_STAMPS (HV_EKF_PHASE_STAMPS=1 in the
// environment of hybvio_amd/build.py): every stamp is a scalar memory wait in the middle of a pipeline.
}  // namespace
__device__ long long g_phase_stamp[16];
namespace {
#ifdef HV_EKF_PHASE_STAMPS
#define PHASE_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_stamp[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PHASE_STAMP(i) do { } while (0)
#endif


__device__ __forceinline__ void normalize4(double *q)
{
    const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nn > 0.0) { q[0] /= nn; q[1] /= nn; q[2] /= nn; q[3] /= nn; }
}

// ---------------------------------------------------------------------------------------------
// predict (ekf.cpp:320-514)
// ---------------------------------------------------------------------------------------------
struct PredictArgs {
    int n, batch;
    double *m, *P, *Q, *dydx;            // per filter: n, n*n, 144, 400
    const double *dt, *gyro, *acc;        // device arrays [batch], [batch*3], [batch*3] (or null -> immediates)
    double dt0, g0[3], a0[3];
    int nsteps;                           // IMU samples per launch; device arrays are [nsteps][batch](x3)
    double noise_scale, gravity, baa, baa_rev, bga, bga_rev;
};

__device__ void quat2rmat_d(const double q[4], double R[9], double dR[36])   // util.cpp:10-47, column-major
{
    const double rows[4][9] = {
        { 2*q[0], -2*q[3],  2*q[2],   2*q[3],  2*q[0], -2*q[1],  -2*q[2],  2*q[1],  2*q[0] },
        { 2*q[1],  2*q[2],  2*q[3],   2*q[2], -2*q[1], -2*q[0],   2*q[3],  2*q[0], -2*q[1] },
        {-2*q[2],  2*q[1],  2*q[0],   2*q[1],  2*q[2],  2*q[3],  -2*q[0],  2*q[3], -2*q[2] },
        {-2*q[3], -2*q[0],  2*q[1],   2*q[0], -2*q[3],  2*q[2],   2*q[1],  2*q[2],  2*q[3] } };
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) dR[9 * k + 3 * j + i] = rows[k][3 * i + j];
    const double Rr[9] = {
        q[0]*q[0]+q[1]*q[1]-q[2]*q[2]-q[3]*q[3], 2*q[1]*q[2] - 2*q[0]*q[3], 2*q[1]*q[3] + 2*q[0]*q[2],
        2*q[1]*q[2] + 2*q[0]*q[3], q[0]*q[0]-q[1]*q[1]+q[2]*q[2]-q[3]*q[3], 2*q[2]*q[3] - 2*q[0]*q[1],
        2*q[1]*q[3] - 2*q[0]*q[2], 2*q[2]*q[3] + 2*q[0]*q[1], q[0]*q[0]-q[1]*q[1]-q[2]*q[2]+q[3]*q[3] };
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[3 * j + i] = Rr[3 * i + j];
}

// cos(th) and sin(th) / th of the half rotation angle th = |w| dt / 2 >= 0 of one IMU sample (ekf.cpp:415-425). Below 0.25 rad -- 100 rad/s at
// dt = 5 ms; every sample a camera rig produces -- both are Taylor polynomials in th^2 up to th^16 / th^17 (remainder < 1e-26), two Horner
// chains side by side on the one lane that owns the mean's serial section, instead of the device library's cos() and sin() with their
// argument reduction and a division (~1.1 k of a sample's ~2.9 k cycles, late r06). Within an ulp of the library's values; the
// reference's own branch for th <= 1e-8 is kept as it is.
__device__ __forceinline__ void cos_sinc(double th, double &c, double &sc)
{
    if (th < 0.25) {
        const double x = th * th;
        double pc = 1.0 / 20922789888000.0, ps = -1.0 / 355687428096000.0;     // 1 / 16!, -1 / 17!
        pc = pc * x - 1.0 / 87178291200.0;   ps = ps * x + 1.0 / 1307674368000.0;   // -1 / 14!, 1 / 15!
        pc = pc * x + 1.0 / 479001600.0;     ps = ps * x - 1.0 / 6227020800.0;      //  1 / 12!, -1 / 13!
        pc = pc * x - 1.0 / 3628800.0;       ps = ps * x + 1.0 / 39916800.0;        // -1 / 10!, 1 / 11!
        pc = pc * x + 1.0 / 40320.0;         ps = ps * x - 1.0 / 362880.0;          //  1 / 8!, -1 / 9!
        pc = pc * x - 1.0 / 720.0;           ps = ps * x + 1.0 / 5040.0;            // -1 / 6!, 1 / 7!
        pc = pc * x + 1.0 / 24.0;            ps = ps * x - 1.0 / 120.0;             //  1 / 4!, -1 / 5!
        pc = pc * x - 0.5;                   ps = ps * x + 1.0 / 6.0;               // -1 / 2!, 1 / 3!
        c = pc * x + 1.0;
        sc = th > 1e-8 ? 1.0 - ps * x : 1.0 - th * th / 6.0;
    } else {
        c = cos(th); sc = sin(th) / th;
    }
}

// entry t = 3 i + j of Rr(q) (util.cpp:10-47), for the lane that owns it: two code paths -- the diagonal's four squares, the off-diagonal's
// two products -- with the operands selected per lane (a `switch` over the nine entries ran its cases one after the other: ~890 cycles of a
// sample's mean recursion, late r06). Shared by both predict kernels, so that they contract to the same FMAs.
__device__ __forceinline__ double quat_rotation_entry(int t, const double *sqn)
{
    const double qq[4] = {sqn[0], sqn[1], sqn[2], sqn[3]};
    const int i = t / 3, j = t - 3 * i;
    if (i == j) {                                           // q0^2 + s1 q1^2 + s2 q2^2 + s3 q3^2, signs (+ - -), (- + -), (- - +)
        const double a1 = qq[1] * qq[1], a2 = qq[2] * qq[2], a3 = qq[3] * qq[3];
        double acc = qq[0] * qq[0];
        acc = i == 0 ? acc + a1 : acc - a1;
        acc = i == 1 ? acc + a2 : acc - a2;
        acc = i == 2 ? acc + a3 : acc - a3;
        return acc;
    }
    // Rr: (0,1) 2q1q2 - 2q0q3, (0,2) 2q1q3 + 2q0q2, (1,0) 2q1q2 + 2q0q3, (1,2) 2q2q3 - 2q0q1, (2,0) 2q1q3 - 2q0q2, (2,1) 2q2q3 + 2q0q1
    const int ia = i + 1, ib = j + 1, ic = 6 - ia - ib;
    const int lo = ia < ib ? ia : ib, hi = ia < ib ? ib : ia;
    const double p = 2 * qq[lo] * qq[hi], r = 2 * qq[0] * qq[ic];
    const bool plus = (i == 0 && j == 2) || (i == 1 && j == 0) || (i == 2 && j == 1);
    return plus ? p + r : p - r;
}

#define F_(i, j) F[(j) * INER + (i)]
#define L_(i, j) Lm[(j) * INER + (i)]

// (two workgroups per CU: 256 registers per lane)
__global__ __launch_bounds__(256, 2) void ekf_predict_kernel(PredictArgs a)
{
    __shared__ double F[INER * INER], Lm[INER * QD], Qs[QD * QD], LQ[INER * QD], P00[INER * INER], FP[INER * INER];
    __shared__ double Phi[INER * INER], PhiN[INER * INER];
    __shared__ double ms[INER], sh[112];           // the inertial part of the mean; stage values of the mean / F / L section
    const int b = blockIdx.x, t = threadIdx.x, n = a.n;
    double *m = a.m + (size_t)b * n, *P = a.P + (size_t)b * n * n, *Q = a.Q + (size_t)b * QD * QD;
    // nsteps IMU samples in one launch: the mean / Jacobian chain and the 20 x 20 block recursion
    // P00 <- F P00 F' + L Q L' run per sample in LDS; the off-diagonal blocks only ever see the product
    // Phi = F_n ... F_1 (P10 <- P10 Phi', P01 <- Phi P01), so they are touched once at the end.
    bool any = false;
    for (int s = 0; s < a.nsteps; s++) { const double d = a.dt ? a.dt[(size_t)s * a.batch + b] : a.dt0; any = any || d > 0.0; }
    if (!any) return;                              // ekf.cpp:365-368 (the host adapter keeps the clock)

    PHASE_STAMP(12);
    // Off-diagonal blocks P10 <- P10 F', P01 <- F P01 (ekf.cpp:506-508) as MFMA items: item I < tiles is
    // the 16-row tile I of P10, item tiles + C the 16-column tile C of P01; either one needs a 16 x 20
    // slab of P (5 k-steps, one f64 per lane each) and all of F. The slabs of the first five items of
    // every wave are requested here, before thread 0 starts on the mean / F / L, so that their HBM
    // round trip hides behind that serial section.
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, kq = lane >> 4, cl = lane & 15;
    const int tiles = (n - INER + 15) >> 4;
    constexpr int NIT = 5;
    auto slab_ptr = [&](int it, int sx) -> double * {
        const int k = min(4 * sx + kq, INER - 1);
        if (it < tiles) return P + (size_t)k * n + min(INER + 16 * it + cl, n - 1);          // P10(i, k): lanes along i
        return P + (size_t)min(INER + 16 * (it - tiles) + cl, n - 1) * n + k;                // P01(k, c): lanes along c
    };
    double slab[NIT][5];
#pragma unroll
    for (int u = 0; u < NIT; u++) {
        const int it = wave + 4 * u;
        if (it < 2 * tiles) {
#pragma unroll
            for (int sx = 0; sx < 5; sx++) slab[u][sx] = *slab_ptr(it, sx);
        }
    }
    for (int i = t; i < INER * INER; i += 256) { Phi[i] = (i % (INER + 1) == 0) ? 1.0 : 0.0; P00[i] = P[(size_t)(i / INER) * n + (i % INER)]; }
    for (int i = t; i < QD * QD; i += 256) Qs[i] = Q[i];
    if (t < INER) ms[t] = m[t];

    double exp_dt = -1.0, e_baa2 = 1.0, e_bga2 = 1.0, e_baa1 = 1.0, e_bga1 = 1.0;     // thread 0 only
    for (int step = 0; step < a.nsteps; step++) {
    const size_t sb = (size_t)step * a.batch + b;
    const double dt = a.dt ? a.dt[sb] : a.dt0;
    if (!(dt > 0.0)) continue;                     // uniform per workgroup
    __syncthreads();
    for (int i = t; i < INER * INER; i += 256) F[i] = (i % (INER + 1) == 0) ? 1.0 : 0.0;
    for (int i = t; i < INER * QD; i += 256) Lm[i] = 0.0;
    __syncthreads();

    // ---- mean, F and L of this sample (ekf.cpp:370-503). r01 ran this section on thread 0 (~1.5 k dependent f64 instructions,
    // half of a sample's time); here every ELEMENT is still evaluated by one lane with the reference's expression and summation
    // order -- results are unchanged -- but the elements of a stage are spread over the lanes of the workgroup, with a barrier
    // where a stage reads what the previous one wrote. Stage scalars and small matrices live in `sh`.
    double *const sA = sh, *const sSrow = sh + 16, *const sqn = sh + 32, *const sprevQ = sh + 36, *const sR = sh + 40,
           *const sdR = sh + 49, *const sTxab = sh + 85, *const sT34 = sh + 88, *const sxa = sh + 100, *const ssc = sh + 103;
    if (t == 0) {
        double xg[3];
        for (int i = 0; i < 3; i++) { xg[i] = a.gyro ? a.gyro[3 * sb + i] : a.g0[i]; sxa[i] = a.acc ? a.acc[3 * sb + i] : a.a0[i]; }
        // the four exponentials of a sample depend on dt only: evaluated when dt changes (IMU samples are equally spaced as a
        // rule, so once per launch), the same values as the per-sample evaluation
        if (dt != exp_dt) {
            exp_dt = dt;
            e_baa2 = a.baa_rev > 0.0 ? (1 - exp(-2 * dt * a.baa_rev)) / (2 * a.baa_rev) : 1.0;
            e_bga2 = a.bga_rev > 0.0 ? (1 - exp(-2 * dt * a.bga_rev)) / (2 * a.bga_rev) : 1.0;
            e_baa1 = exp(-dt * a.baa_rev); e_bga1 = exp(-dt * a.bga_rev);
        }
        ssc[2] = e_baa1; ssc[3] = e_bga1;
        if (a.baa > 0.0) {                          // ekf.cpp:397-404
            double v = a.noise_scale * a.baa * a.baa;
            if (a.baa_rev > 0.0) v *= e_baa2;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Qs[(Q_BAA_DRIFT + j) * QD + Q_BAA_DRIFT + i] = (i == j) ? v : 0.0;
        }
        if (a.bga > 0.0) {                          // ekf.cpp:405-412
            double v = a.noise_scale * a.bga * a.bga;
            if (a.bga_rev > 0.0) v *= e_bga2;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Qs[(Q_BGA_DRIFT + j) * QD + Q_BGA_DRIFT + i] = (i == j) ? v : 0.0;
        }
        // A = exp(-dt/2 Omega(w)) = cos(th) I + sin(th)/th S, th = |w| dt/2   (ekf.cpp:415-425)
        const double w[3] = { xg[0] - ms[BGA], xg[1] - ms[BGA + 1], xg[2] - ms[BGA + 2] };
        const double Srow[16] = { 0, -w[0], -w[1], -w[2],  w[0], 0, -w[2], w[1],  w[1], w[2], 0, -w[0],  w[2], -w[1], w[0], 0 };
#pragma unroll
        for (int i = 0; i < 16; i++) sSrow[i] = Srow[i];
        const double th = sqrt(w[0]*w[0] + w[1]*w[1] + w[2]*w[2]) * dt / 2;
        cos_sinc(th, ssc[0], ssc[1]);
    }
    __syncthreads();
    {   // stage B: A (column-major 4x4), the previous quaternion, T xa - ba, the position
        const double c = ssc[0], sc = ssc[1];
        if (t < 16) { const int i = t & 3, j = t >> 2; sA[4 * j + i] = sc * sSrow[4 * i + j] * (-dt / 2) + (i == j ? c : 0.0); }
        else if (t < 20) sprevQ[t - 16] = ms[ORI + t - 16];
        else if (t >= 32 && t < 35) { const int i = t - 32; sTxab[i] = ms[BAT + i] * sxa[i] - ms[BAA + i]; }
        else if (t >= 40 && t < 43) { const int i = t - 40; ms[POS + i] += ms[VEL + i] * dt; }
    }
    __syncthreads();
    if (t < 4) { double s_ = 0; for (int j = 0; j < 4; j++) s_ += sA[4 * j + t] * sprevQ[j]; sqn[t] = s_; }
    __syncthreads();
    {   // stage D: R(q) and dR/dq (util.cpp:10-47, column-major): one element per lane
        const double q0 = sqn[0], q1 = sqn[1], q2 = sqn[2], q3 = sqn[3];
        if (t < 9) {
            const int i = t / 3, j = t - 3 * i;                 // Rr[3 i + j] -> R[3 j + i]
            const double v = quat_rotation_entry(t, sqn);
            sR[3 * j + i] = v;
        } else if (t >= 64 && t < 100) {
            // rows[k][3 i + j] = sign * 2 q[idx]: two bits of index and one of sign per entry
            const int e = t - 64, k = e / 9, ij = e - 9 * k, i = ij / 3, j = ij - 3 * i;
            constexpr unsigned char IDX[4][9] = { {0,3,2, 3,0,1, 2,1,0}, {1,2,3, 2,1,0, 3,0,1}, {2,1,0, 1,2,3, 0,3,2}, {3,0,1, 0,3,2, 1,2,3} };
            constexpr signed char SGN[4][9] = { {1,-1,1, 1,1,-1, -1,1,1}, {1,1,1, 1,-1,-1, 1,1,-1}, {-1,1,1, 1,1,1, -1,1,-1}, {-1,-1,1, 1,-1,1, 1,1,1} };
            const int id = IDX[k][ij];
            const double qv = id == 0 ? q0 : id == 1 ? q1 : id == 2 ? q2 : q3;
            sdR[9 * k + 3 * j + i] = SGN[k][ij] > 0 ? 2 * qv : -2 * qv;
        }
    }
    __syncthreads();
    {   // stage E: the rest of the mean; the entries of F and L that depend on R, dR, A only
        if (t < 3) {
            const int i = t;
            const double grav = i == 2 ? -a.gravity : 0.0;
            double s_ = 0; for (int j = 0; j < 3; j++) s_ += sR[3 * i + j] * sTxab[j];
            ms[VEL + i] += (s_ + grav) * dt;
            F_(POS + i, VEL + i) = dt;
            L_(BGA + i, Q_BGA_DRIFT + i) = 1.0; L_(BAA + i, Q_BAA_DRIFT + i) = 1.0;
        } else if (t >= 4 && t < 8) ms[ORI + t - 4] = sqn[t - 4];
        else if (t >= 8 && t < 11) { if (a.baa > 0.0) ms[BAA + t - 8] *= ssc[2]; }
        else if (t >= 12 && t < 15) { if (a.bga > 0.0) ms[BGA + t - 12] *= ssc[3]; }
        else if (t >= 16 && t < 28) {
            const int e = t - 16, k = e / 3, i = e - 3 * k;
            double s_ = 0; for (int j = 0; j < 3; j++) s_ += sdR[9 * k + 3 * i + j] * sTxab[j];
            sT34[3 * k + i] = s_ * dt;
        } else if (t >= 32 && t < 44) {
            const int e = t - 32, g = e >> 2, i = e & 3;
            const double h = dt / 2;
            const double dS[3][16] = {
                { 0, h, 0, 0,  -h, 0, 0, 0,  0, 0, 0, h,  0, 0, -h, 0 },
                { 0, 0, h, 0,  0, 0, 0, -h,  -h, 0, 0, 0,  0, h, 0, 0 },
                { 0, 0, 0, h,  0, 0, h, 0,  0, -h, 0, 0,  -h, 0, 0, 0 } };
            double t1[4];
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                double s_ = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { const double d = g == 0 ? dS[0][4 * ii + j] : g == 1 ? dS[1][4 * ii + j] : dS[2][4 * ii + j]; s_ += d * sprevQ[j]; }
                t1[ii] = s_;
            }
            double s_ = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) s_ += sA[4 * j + i] * t1[j];
            L_(ORI + i, Q_GYRO + g) = s_;
        } else if (t >= 48 && t < 57) {
            const int e = t - 48, i = e / 3, j = e - 3 * i;
            L_(VEL + i, Q_ACC + j) = sR[3 * i + j] * dt;
            F_(VEL + i, BAA + j) = -sR[3 * i + j] * dt; F_(VEL + i, BAT + j) = sR[3 * i + j] * sxa[j] * dt;
        } else if (t >= 64 && t < 80) {
            const int e = t - 64, i = e & 3, j = e >> 2;
            F_(ORI + i, ORI + j) = sA[4 * j + i];
        }
    }
    __syncthreads();
    if (t < 12) {
        const int i = t / 4, j = t - 4 * i;
        double s_ = 0; for (int k = 0; k < 4; k++) s_ += sT34[3 * k + i] * sA[4 * j + k];
        F_(VEL + i, ORI + j) = s_;
    }
    __syncthreads();
    if (t < 9) {
        const int i = t / 3, g = t - 3 * i;
        double s_ = 0; for (int k = 0; k < 4; k++) s_ += F_(VEL + i, ORI + k) * L_(ORI + k, Q_GYRO + g);
        L_(VEL + i, Q_GYRO + g) = s_;
        F_(VEL + i, BGA + g) = -s_;
    } else if (t >= 16 && t < 28) {
        const int e = t - 16, i = e / 3, g = e - 3 * i;
        F_(ORI + i, BGA + g) = -L_(ORI + i, Q_GYRO + g);
    }
    __syncthreads();

    PHASE_STAMP(13);
    // P00 = F P00 F' + L Q L' (ekf.cpp:504-505) in LDS; Phi <- F Phi. The 20 x 20 results are four 16 x 16 MFMA tiles, one per
    // wavefront (f64 16x16x4, 5 k-steps): a chain of 5 MFMAs per product where the r01 loops spent 20-32 dependent LDS round trips per
    // element (a sample's products 3 x ~1.3 us -> well under 1 us; the summation order inside a product changes, ~1e-16 relative)
    {
        const int ti = wave & 1, tj = wave >> 1;                 // output rows 16 ti .., columns 16 tj ..
        const int mi = ti ? INER - 16 : 16, nj = tj ? INER - 16 : 16;
        // A(i, k) = F_(16 ti + i, k) = F[k INER + 16 ti + i];  B(k, j) = P00(k, 16 tj + j) = P00[(16 tj + j) INER + k]
        const double4v fp = mfma_tile(F + 16 * ti, 1, INER, mi, P00 + 16 * tj * INER, 1, INER, nj, INER);
        const double4v ph = mfma_tile(F + 16 * ti, 1, INER, mi, Phi + 16 * tj * INER, 1, INER, nj, INER);
        // L Q (20 x 12): row tiles on wavefronts 0 and 1. LQ(i, j) = sum_k L(i, k) Qs(j, k): B(k, j) = Qs[j QD + k]
        double4v lq = {0.0, 0.0, 0.0, 0.0};
        if (wave < 2) lq = mfma_tile(Lm + 16 * wave, 1, INER, wave ? INER - 16 : 16, Qs, 1, QD, QD, QD);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = kq + 4 * q;
            if (r < mi && cl < nj) { FP[(16 * tj + cl) * INER + 16 * ti + r] = fp[q]; PhiN[(16 * tj + cl) * INER + 16 * ti + r] = ph[q]; }
            if (wave < 2 && r < (wave ? INER - 16 : 16) && cl < QD) LQ[cl * INER + 16 * wave + r] = lq[q];
        }
        __syncthreads();
        // P00(i, j) = sum_k FP(i, k) F_(j, k) + sum_k LQ(i, k) L_(j, k): B(k, j) = F[k INER + 16 tj + j] resp. Lm[k INER + 16 tj + j]
        const double4v p1 = mfma_tile(FP + 16 * ti, 1, INER, mi, F + 16 * tj, INER, 1, nj, INER);
        const double4v p2 = mfma_tile(LQ + 16 * ti, 1, INER, mi, Lm + 16 * tj, INER, 1, nj, QD);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = kq + 4 * q;
            if (r < mi && cl < nj) P00[(16 * tj + cl) * INER + 16 * ti + r] = p1[q] + p2[q];
        }
        for (int e = t; e < INER * INER; e += 256) Phi[e] = PhiN[e];
    }
    }
    __syncthreads();
    for (int i = t; i < INER * INER; i += 256) { P[(size_t)(i / INER) * n + (i % INER)] = P00[i]; a.dydx[(size_t)b * INER * INER + i] = F[i]; }
    for (int i = t; i < QD * QD; i += 256) Q[i] = Qs[i];
    if (t < INER) m[t] = ms[t];
    PHASE_STAMP(14);
    {
        // F operand (the product Phi of the steps' F), shared by every item: lane (kq, cl) holds Phi(cl + 16 tt, 4 s + kq). It is the A operand
        // of (P10 F')' = F P10' (rows c of F, columns i of the tile) and the B operand of (F P01)' =
        // P01' F' (rows c of the tile, columns r of F): both products are formed transposed so that the
        // 16 lanes of an output row group store 16 consecutive doubles.
        double f0[5], f1[5];
#pragma unroll
        for (int sx = 0; sx < 5; sx++) {
            const int k = 4 * sx + kq;                                                        // < 20
            f0[sx] = Phi[k * INER + cl];
            f1[sx] = Phi[k * INER + min(16 + cl, INER - 1)];
        }
        for (int base = 0; base < 2 * tiles; base += 4 * NIT) {
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                const int it = base + wave + 4 * u;
                if (it < 2 * tiles) {
                    if (base > 0) {
#pragma unroll
                        for (int sx = 0; sx < 5; sx++) slab[u][sx] = *slab_ptr(it, sx);
                    }
                    double4v a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
                    const bool p10 = it < tiles;
#pragma unroll
                    for (int sx = 0; sx < 5; sx++) {
                        const double x0 = p10 ? f0[sx] : slab[u][sx], y0 = p10 ? slab[u][sx] : f0[sx];
                        const double x1 = p10 ? f1[sx] : slab[u][sx], y1 = p10 ? slab[u][sx] : f1[sx];
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int rr = kq + 4 * q;                                            // output row within the tile
                        if (p10) {
                            const int i = INER + 16 * it + cl;                                // (c = rr [+16], i)
                            if (i < n) {
                                P[(size_t)rr * n + i] = a0[q];
                                if (16 + rr < INER) P[(size_t)(16 + rr) * n + i] = a1[q];
                            }
                        } else {
                            const int c = INER + 16 * (it - tiles) + rr;                      // (c, r = cl [+16])
                            if (c < n) {
                                P[(size_t)c * n + cl] = a0[q];
                                if (16 + cl < INER) P[(size_t)c * n + 16 + cl] = a1[q];
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    PHASE_STAMP(15);
}


// ---------------------------------------------------------------------------------------------
// ekf_predict_chain_kernel (late r06): the same arithmetic as ekf_predict_kernel, element by element and in the same
// order -- results are bit-identical --, with the sample-to-sample dependency reduced to what really depends: the MEAN. A sample's F and L
// only read quantities of the mean recursion (A, the old and the new quaternion, R, T xa - ba, dt), so
//   phase 1  ONE wavefront runs the mean recursion of up to PCH samples back to back -- five short stages per sample that talk through LDS
//            inside the wave (DS operations of a wave execute in order: a wave-level fence, no workgroup barrier) -- and leaves every
//            sample's stage values in its own slot;
//   phase 2  all 256 threads form dR, F and L of ALL those samples at once (four workgroup barriers per chunk instead of nine per sample);
//   phase 3  the 20 x 20 recursions P00 <- F P00 F' + L Q L', Phi <- F Phi sample by sample (two barriers each), as before.
// ekf_predict_kernel spent ~5.3 us per sample on nine barrier-separated stages (profiles/r06/predict_pipe_ab.txt); knob ekf_predict_chain 0
// keeps it.
constexpr int PCH = 5;                  // samples per chunk (F and L of a chunk stay in LDS: 5 x (400 + 240) doubles)
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256, 2) void ekf_predict_chain_kernel(PredictArgs a)
{
    constexpr int SHW = 112;                                  // stage values of one sample (the layout of ekf_predict_kernel's `sh`) ...
    constexpr int X_DT = 107, X_LIVE = 108, X_VBAA = 109, X_VBGA = 110;   // ... + its dt, whether it runs, the drift variances of Q
    __shared__ double Fs[PCH * INER * INER], Ls[PCH * INER * QD], Qs[QD * QD], LQ[INER * QD], P00[INER * INER], FP[INER * INER];
    __shared__ double Phi[INER * INER], PhiN[INER * INER];
    __shared__ double ms[INER], shs[PCH * SHW], imu[PCH * 8];    // imu: dt, gyro[3], acc[3] of the chunk's samples
    const int b = blockIdx.x, t = threadIdx.x, n = a.n;
    double *m = a.m + (size_t)b * n, *P = a.P + (size_t)b * n * n, *Q = a.Q + (size_t)b * QD * QD;
    bool any = false;
    for (int s = 0; s < a.nsteps; s++) { const double d = a.dt ? a.dt[(size_t)s * a.batch + b] : a.dt0; any = any || d > 0.0; }
    if (!any) return;                              // ekf.cpp:365-368 (the host adapter keeps the clock)

    PHASE_STAMP(12);
    // off-diagonal items: the slabs of the first five items of every wave are requested here (see ekf_predict_kernel)
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, kq = lane >> 4, cl = lane & 15;
    const int tiles = (n - INER + 15) >> 4;
    constexpr int NIT = 5;
    auto slab_ptr = [&](int it, int sx) -> double * {
        const int k = min(4 * sx + kq, INER - 1);
        if (it < tiles) return P + (size_t)k * n + min(INER + 16 * it + cl, n - 1);          // P10(i, k): lanes along i
        return P + (size_t)min(INER + 16 * (it - tiles) + cl, n - 1) * n + k;                // P01(k, c): lanes along c
    };
    double slab[NIT][5];
#pragma unroll
    for (int u = 0; u < NIT; u++) {
        const int it = wave + 4 * u;
        if (it < 2 * tiles) {
#pragma unroll
            for (int sx = 0; sx < 5; sx++) slab[u][sx] = *slab_ptr(it, sx);
        }
    }
    for (int i = t; i < INER * INER; i += 256) { Phi[i] = (i % (INER + 1) == 0) ? 1.0 : 0.0; P00[i] = P[(size_t)(i / INER) * n + (i % INER)]; }
    for (int i = t; i < QD * QD; i += 256) Qs[i] = Q[i];
    if (t < INER) ms[t] = m[t];
    __syncthreads();

    double exp_dt = -1.0, e_baa2 = 1.0, e_bga2 = 1.0, e_baa1 = 1.0, e_bga1 = 1.0;     // lane 0 of wave 0 only
    for (int s0 = 0; s0 < a.nsteps; s0 += PCH) {
        const int ns = min(PCH, a.nsteps - s0);
        // the chunk's IMU samples in ONE round trip (requested per sample, every sample's chain started with two dependent global loads)
        if (t < ns * 8) {
            const int sl = t >> 3, e = t & 7;
            const size_t sb = (size_t)(s0 + sl) * a.batch + b;
            double v = 0.0;
            if (e == 0) v = a.dt ? a.dt[sb] : a.dt0;
            else if (e < 4) v = a.gyro ? a.gyro[3 * sb + e - 1] : a.g0[e - 1];
            else if (e < 7) v = a.acc ? a.acc[3 * sb + e - 4] : a.a0[e - 4];
            imu[t] = v;
        }
        __syncthreads();
        // ---- phase 1: the mean recursion of the chunk's samples (ekf.cpp:370-503, the entries of the mean), wave 0 ----
        if (wave == 0) {
            for (int sl = 0; sl < ns; sl++) {
                double *const sh = shs + sl * SHW;
                double *const sA = sh, *const sSrow = sh + 16, *const sqn = sh + 32, *const sprevQ = sh + 36, *const sR = sh + 40,
                       *const sTxab = sh + 85, *const sxa = sh + 100, *const ssc = sh + 103;
                const double dt = imu[8 * sl];
                if (t == 0) { sh[X_DT] = dt; sh[X_LIVE] = dt > 0.0 ? 1.0 : 0.0; }
                if (!(dt > 0.0)) continue;                     // uniform
                if (s0 == 0 && sl == 0) PHASE_STAMP(0);
                if (t == 0) {
                    double xg[3];
                    for (int i = 0; i < 3; i++) { xg[i] = imu[8 * sl + 1 + i]; sxa[i] = imu[8 * sl + 4 + i]; }
                    if (dt != exp_dt) {
                        exp_dt = dt;
                        e_baa2 = a.baa_rev > 0.0 ? (1 - exp(-2 * dt * a.baa_rev)) / (2 * a.baa_rev) : 1.0;
                        e_bga2 = a.bga_rev > 0.0 ? (1 - exp(-2 * dt * a.bga_rev)) / (2 * a.bga_rev) : 1.0;
                        e_baa1 = exp(-dt * a.baa_rev); e_bga1 = exp(-dt * a.bga_rev);
                    }
                    ssc[2] = e_baa1; ssc[3] = e_bga1;
                    if (a.baa > 0.0) { double v = a.noise_scale * a.baa * a.baa; if (a.baa_rev > 0.0) v *= e_baa2; sh[X_VBAA] = v; }   // ekf.cpp:397-404
                    if (a.bga > 0.0) { double v = a.noise_scale * a.bga * a.bga; if (a.bga_rev > 0.0) v *= e_bga2; sh[X_VBGA] = v; }   // ekf.cpp:405-412
                    // A = exp(-dt/2 Omega(w)) = cos(th) I + sin(th)/th S, th = |w| dt/2   (ekf.cpp:415-425)
                    const double w[3] = { xg[0] - ms[BGA], xg[1] - ms[BGA + 1], xg[2] - ms[BGA + 2] };
                    const double Srow[16] = { 0, -w[0], -w[1], -w[2],  w[0], 0, -w[2], w[1],  w[1], w[2], 0, -w[0],  w[2], -w[1], w[0], 0 };
#pragma unroll
                    for (int i = 0; i < 16; i++) sSrow[i] = Srow[i];
                    const double th = sqrt(w[0]*w[0] + w[1]*w[1] + w[2]*w[2]) * dt / 2;
                    cos_sinc(th, ssc[0], ssc[1]);
                }
                wave_sync_lds();
                if (s0 == 0 && sl == 0) PHASE_STAMP(1);
                {   // stage B: A (column-major 4x4), the previous quaternion, T xa - ba, the position
                    const double c = ssc[0], sc = ssc[1];
                    if (t < 16) { const int i = t & 3, j = t >> 2; sA[4 * j + i] = sc * sSrow[4 * i + j] * (-dt / 2) + (i == j ? c : 0.0); }
                    else if (t < 20) sprevQ[t - 16] = ms[ORI + t - 16];
                    else if (t >= 32 && t < 35) { const int i = t - 32; sTxab[i] = ms[BAT + i] * sxa[i] - ms[BAA + i]; }
                    else if (t >= 40 && t < 43) { const int i = t - 40; ms[POS + i] += ms[VEL + i] * dt; }
                }
                wave_sync_lds();
                if (s0 == 0 && sl == 0) PHASE_STAMP(2);
                if (t < 4) { double s_ = 0; for (int j = 0; j < 4; j++) s_ += sA[4 * j + t] * sprevQ[j]; sqn[t] = s_; }
                wave_sync_lds();
                if (s0 == 0 && sl == 0) PHASE_STAMP(3);
                if (t < 9) {   // stage D, the part the mean needs: R(q) (util.cpp:10-47, column-major)
                    const int i = t / 3, j = t - 3 * i;                 // Rr[3 i + j] -> R[3 j + i]
                    sR[3 * j + i] = quat_rotation_entry(t, sqn);
                }
                wave_sync_lds();
                if (s0 == 0 && sl == 0) PHASE_STAMP(4);
                {   // stage E, the entries of the mean
                    if (t < 3) {
                        const int i = t;
                        const double grav = i == 2 ? -a.gravity : 0.0;
                        double s_ = 0; for (int j = 0; j < 3; j++) s_ += sR[3 * i + j] * sTxab[j];
                        ms[VEL + i] += (s_ + grav) * dt;
                    } else if (t >= 4 && t < 8) ms[ORI + t - 4] = sqn[t - 4];
                    else if (t >= 8 && t < 11) { if (a.baa > 0.0) ms[BAA + t - 8] *= ssc[2]; }
                    else if (t >= 12 && t < 15) { if (a.bga > 0.0) ms[BGA + t - 12] *= ssc[3]; }
                }
                wave_sync_lds();
                if (s0 == 0 && sl == 0) PHASE_STAMP(5);
            }
        }
        __syncthreads();
        PHASE_STAMP(13);
        // ---- phase 2: dR, F and L of every live sample of the chunk, one element per lane and pass, the reference's expressions ----
        for (int i = t; i < ns * INER * INER; i += 256) Fs[i] = ((i % (INER * INER)) % (INER + 1) == 0) ? 1.0 : 0.0;
        for (int i = t; i < ns * INER * QD; i += 256) Ls[i] = 0.0;
        for (int w = t; w < ns * 36; w += 256) {                        // stage D: dR / dq
            const int sl = w / 36, e = w - 36 * sl;
            double *const sh = shs + sl * SHW;
            if (sh[X_LIVE] == 0.0) continue;
            const double q0 = sh[32], q1 = sh[33], q2 = sh[34], q3 = sh[35];
            const int k = e / 9, ij = e - 9 * k, i = ij / 3, j = ij - 3 * i;
            constexpr unsigned char IDX[4][9] = { {0,3,2, 3,0,1, 2,1,0}, {1,2,3, 2,1,0, 3,0,1}, {2,1,0, 1,2,3, 0,3,2}, {3,0,1, 0,3,2, 1,2,3} };
            constexpr signed char SGN[4][9] = { {1,-1,1, 1,1,-1, -1,1,1}, {1,1,1, 1,-1,-1, 1,1,-1}, {-1,1,1, 1,1,1, -1,1,-1}, {-1,-1,1, 1,-1,1, 1,1,1} };
            const int id = IDX[k][ij];
            const double qv = id == 0 ? q0 : id == 1 ? q1 : id == 2 ? q2 : q3;
            sh[49 + 9 * k + 3 * j + i] = SGN[k][ij] > 0 ? 2 * qv : -2 * qv;
        }
        __syncthreads();
        for (int w = t; w < ns * 64; w += 256) {                        // stage E: the entries of F and L that depend on R, dR, A only
            const int sl = w >> 6, tt = w & 63;
            double *const sh = shs + sl * SHW;
            if (sh[X_LIVE] == 0.0) continue;
            double *const F = Fs + sl * INER * INER, *const Lm = Ls + sl * INER * QD;
            const double *const sA = sh, *const sprevQ = sh + 36, *const sR = sh + 40, *const sdR = sh + 49, *const sTxab = sh + 85, *const sxa = sh + 100;
            double *const sT34 = sh + 88;
            const double dt = sh[X_DT];
            if (tt < 3) {
                const int i = tt;
                F_(POS + i, VEL + i) = dt;
                L_(BGA + i, Q_BGA_DRIFT + i) = 1.0; L_(BAA + i, Q_BAA_DRIFT + i) = 1.0;
            } else if (tt >= 4 && tt < 16) {
                const int e = tt - 4, k = e / 3, i = e - 3 * k;
                double s_ = 0; for (int j = 0; j < 3; j++) s_ += sdR[9 * k + 3 * i + j] * sTxab[j];
                sT34[3 * k + i] = s_ * dt;
            } else if (tt >= 16 && tt < 28) {
                const int e = tt - 16, g = e >> 2, i = e & 3;
                const double h = dt / 2;
                const double dS[3][16] = {
                    { 0, h, 0, 0,  -h, 0, 0, 0,  0, 0, 0, h,  0, 0, -h, 0 },
                    { 0, 0, h, 0,  0, 0, 0, -h,  -h, 0, 0, 0,  0, h, 0, 0 },
                    { 0, 0, 0, h,  0, 0, h, 0,  0, -h, 0, 0,  -h, 0, 0, 0 } };
                double t1[4];
#pragma unroll
                for (int ii = 0; ii < 4; ii++) {
                    double s_ = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) { const double d = g == 0 ? dS[0][4 * ii + j] : g == 1 ? dS[1][4 * ii + j] : dS[2][4 * ii + j]; s_ += d * sprevQ[j]; }
                    t1[ii] = s_;
                }
                double s_ = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) s_ += sA[4 * j + i] * t1[j];
                L_(ORI + i, Q_GYRO + g) = s_;
            } else if (tt >= 28 && tt < 37) {
                const int e = tt - 28, i = e / 3, j = e - 3 * i;
                L_(VEL + i, Q_ACC + j) = sR[3 * i + j] * dt;
                F_(VEL + i, BAA + j) = -sR[3 * i + j] * dt; F_(VEL + i, BAT + j) = sR[3 * i + j] * sxa[j] * dt;
            } else if (tt >= 40 && tt < 56) {
                const int e = tt - 40, i = e & 3, j = e >> 2;
                F_(ORI + i, ORI + j) = sA[4 * j + i];
            }
        }
        __syncthreads();
        for (int w = t; w < ns * 12; w += 256) {
            const int sl = w / 12, tt = w - 12 * sl;
            double *const sh = shs + sl * SHW;
            if (sh[X_LIVE] == 0.0) continue;
            double *const F = Fs + sl * INER * INER;
            const double *const sA = sh, *const sT34 = sh + 88;
            const int i = tt / 4, j = tt - 4 * i;
            double s_ = 0; for (int k = 0; k < 4; k++) s_ += sT34[3 * k + i] * sA[4 * j + k];
            F_(VEL + i, ORI + j) = s_;
        }
        __syncthreads();
        for (int w = t; w < ns * 32; w += 256) {
            const int sl = w >> 5, tt = w & 31;
            double *const sh = shs + sl * SHW;
            if (sh[X_LIVE] == 0.0) continue;
            double *const F = Fs + sl * INER * INER, *const Lm = Ls + sl * INER * QD;
            if (tt < 9) {
                const int i = tt / 3, g = tt - 3 * i;
                double s_ = 0; for (int k = 0; k < 4; k++) s_ += F_(VEL + i, ORI + k) * L_(ORI + k, Q_GYRO + g);
                L_(VEL + i, Q_GYRO + g) = s_;
                F_(VEL + i, BGA + g) = -s_;
            } else if (tt >= 16 && tt < 28) {
                const int e = tt - 16, i = e / 3, g = e - 3 * i;
                F_(ORI + i, BGA + g) = -L_(ORI + i, Q_GYRO + g);
            }
        }
        if (s0 == 0) PHASE_STAMP(6);
        // ---- phase 3: P00 = F P00 F' + L Q L' (ekf.cpp:504-505), Phi <- F Phi, sample by sample (see ekf_predict_kernel) ----
        int last_live = -1;
        bool q_set = false;
        // the drift variances of Q (ekf.cpp:397-412) of one sample: Qs is read by the first product phase of a sample only
        auto set_q = [&](const double *sh_q) {
            if (a.baa > 0.0 && t < 9) { const int i = t % 3, j = t / 3; Qs[(Q_BAA_DRIFT + j) * QD + Q_BAA_DRIFT + i] = (i == j) ? sh_q[X_VBAA] : 0.0; }
            if (a.bga > 0.0 && t >= 16 && t < 25) { const int e = t - 16, i = e % 3, j = e / 3; Qs[(Q_BGA_DRIFT + j) * QD + Q_BGA_DRIFT + i] = (i == j) ? sh_q[X_VBGA] : 0.0; }
        };
        for (int sl = 0; sl < ns; sl++) {
            double *const sh = shs + sl * SHW;
            __syncthreads();                                     // (F, L of the chunk; the previous sample's P00 and Phi)
            if (sh[X_LIVE] == 0.0) continue;                     // uniform
            last_live = sl;
            const double *const F = Fs + sl * INER * INER, *const Lm = Ls + sl * INER * QD;
            if (!q_set) { set_q(sh); __syncthreads(); }           // (the chunk's first live sample; the others: behind the previous one's middle barrier)
            q_set = false;
            if (s0 == 0 && sl == 1) PHASE_STAMP(7);
            const int ti = wave & 1, tj = wave >> 1;                 // output rows 16 ti .., columns 16 tj ..
            const int mi = ti ? INER - 16 : 16, nj = tj ? INER - 16 : 16;
            const double4v fp = mfma_tile<5>(F + 16 * ti, 1, INER, mi, P00 + 16 * tj * INER, 1, INER, nj, INER);
            const double4v ph = mfma_tile<5>(F + 16 * ti, 1, INER, mi, Phi + 16 * tj * INER, 1, INER, nj, INER);
            double4v lq = {0.0, 0.0, 0.0, 0.0};
            if (wave < 2) lq = mfma_tile<3>(Lm + 16 * wave, 1, INER, wave ? INER - 16 : 16, Qs, 1, QD, QD, QD);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = kq + 4 * q;
                if (r < mi && cl < nj) { FP[(16 * tj + cl) * INER + 16 * ti + r] = fp[q]; PhiN[(16 * tj + cl) * INER + 16 * ti + r] = ph[q]; }
                if (wave < 2 && r < (wave ? INER - 16 : 16) && cl < QD) LQ[cl * INER + 16 * wave + r] = lq[q];
            }
            if (s0 == 0 && sl == 1) PHASE_STAMP(8);
            __syncthreads();
            if (s0 == 0 && sl == 1) PHASE_STAMP(9);
            {   // the next live sample's Q entries (uniform search)
                int nx = sl + 1;
                while (nx < ns && shs[nx * SHW + X_LIVE] == 0.0) nx++;
                if (nx < ns) { set_q(shs + nx * SHW); q_set = true; }
            }
            const double4v p1 = mfma_tile<5>(FP + 16 * ti, 1, INER, mi, F + 16 * tj, INER, 1, nj, INER);
            const double4v p2 = mfma_tile<3>(LQ + 16 * ti, 1, INER, mi, Lm + 16 * tj, INER, 1, nj, QD);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = kq + 4 * q;
                if (r < mi && cl < nj) P00[(16 * tj + cl) * INER + 16 * ti + r] = p1[q] + p2[q];
            }
            for (int e = t; e < INER * INER; e += 256) Phi[e] = PhiN[e];
            if (s0 == 0 && sl == 1) PHASE_STAMP(10);
        }
        __syncthreads();
        if (s0 == 0) PHASE_STAMP(11);
        if (last_live >= 0)                                       // der_predict reads the LAST sample's F (ekf_predict_kernel: F after its loop)
            for (int i = t; i < INER * INER; i += 256) a.dydx[(size_t)b * INER * INER + i] = Fs[last_live * INER * INER + i];
        __syncthreads();                                          // (the next chunk overwrites the stage values, F and L)
    }
    for (int i = t; i < INER * INER; i += 256) P[(size_t)(i / INER) * n + (i % INER)] = P00[i];
    for (int i = t; i < QD * QD; i += 256) Q[i] = Qs[i];
    if (t < INER) m[t] = ms[t];
    PHASE_STAMP(14);
    {
        double f0[5], f1[5];
#pragma unroll
        for (int sx = 0; sx < 5; sx++) {
            const int k = 4 * sx + kq;                                                        // < 20
            f0[sx] = Phi[k * INER + cl];
            f1[sx] = Phi[k * INER + min(16 + cl, INER - 1)];
        }
        for (int base = 0; base < 2 * tiles; base += 4 * NIT) {
#pragma unroll
            for (int u = 0; u < NIT; u++) {
                const int it = base + wave + 4 * u;
                if (it < 2 * tiles) {
                    if (base > 0) {
#pragma unroll
                        for (int sx = 0; sx < 5; sx++) slab[u][sx] = *slab_ptr(it, sx);
                    }
                    double4v a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
                    const bool p10 = it < tiles;
#pragma unroll
                    for (int sx = 0; sx < 5; sx++) {
                        const double x0 = p10 ? f0[sx] : slab[u][sx], y0 = p10 ? slab[u][sx] : f0[sx];
                        const double x1 = p10 ? f1[sx] : slab[u][sx], y1 = p10 ? slab[u][sx] : f1[sx];
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int rr = kq + 4 * q;                                            // output row within the tile
                        if (p10) {
                            const int i = INER + 16 * it + cl;                                // (c = rr [+16], i)
                            if (i < n) {
                                P[(size_t)rr * n + i] = a0[q];
                                if (16 + rr < INER) P[(size_t)(16 + rr) * n + i] = a1[q];
                            }
                        } else {
                            const int c = INER + 16 * (it - tiles) + rr;                      // (c, r = cl [+16])
                            if (c < n) {
                                P[(size_t)c * n + cl] = a0[q];
                                if (16 + cl < INER) P[(size_t)c * n + 16 + cl] = a1[q];
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    PHASE_STAMP(15);
}

// ---------------------------------------------------------------------------------------------
// update / gate (ekf.cpp:57-82, 760-844)
// ---------------------------------------------------------------------------------------------
struct UpdateArgs {
    int n, nr, l, R, Rs;              // R = nr + n + 1 rows of the tall matrix; Rs = its column stride
    int mode;                         // 0 gate only, 1 update, 2 update only if the chi2 gate passes
    int generic;                      // residual = y - H m[0:l] (ekf.cpp:76-79) instead of the given v
    int normalize_all, use_lds, map_dim;
    double *m, *P;
    const double *H, *v;              // per filter: nr*l column-major, nr
    const double *rdiag;              // per filter diagonal of R (already scaled), or null -> rd0
    double rd0, noise_scale;
    double rd1;                       // mode 3 only: R of the update (rd0 is then the R of the gate)
    double *ws;                       // per filter R*nr doubles (used when the tall matrix exceeds LDS)
    double *chi2; int *status;        // optional outputs
    const unsigned char *active;      // optional per-filter enable
    const int *require_inlier;        // optional per-filter gate result of an earlier launch: run only where it is 0 (INLIER)
    int *success_counter;             // optional per-filter count of applied visual updates (updateSuccessCount, backend.cpp:1183)
    // speculative frame loop (small batches, hv_ekf_visual_frame_dev): inputs / outputs are [n_tracks][batch] records
    //   spec 1: grid (batch, n_tracks), chi2 gate of EVERY pending track (index >= cursor[filter]) against the current (m, P)
    //   spec 2: grid (batch): the first pending track whose gate said INLIER is applied (mode 1), cursor moves behind it
    //   spec 3: spec 1 and spec 2 in ONE launch (mode 3, grid (batch, n_tracks)): every pending track is gated; an inlier then waits
    //           for the gate results of the pending tracks in front of it (they belong to workgroups with a smaller linear index:
    //           dispatched earlier, so the wait cannot deadlock) and, if none of them is an inlier, carries on into the update with
    //           the H P it already holds. The workgroup of the last track closes the pass when nobody applied anything.
    int spec, n_tracks, max_successful;
    int *cursor;                      // [batch] first track of the filter that is not final yet
    const int *gate_in;               // spec 2: [n_tracks][batch] gate results of spec 1 (a.status stays free for this launch's own output)
    const int *nr_rec;                // ragged batches: rows of every record (<= nr, which is then the record stride of H and v; 0: none)
    int *cursor_out;                  // spec 3: the cursor after this pass (ping-pong: late workgroups still read the old one)
    int *pub;                         // spec 3: [n_tracks][batch] published gate decisions, pass_id * 4 + {1 not applicable, 2 inlier}
    int pass_id;                      // spec 3: > 0, distinct per pass of a frame (pub is zeroed per frame)
    int *err;                         // spec 3: device error word of the filter batch (bit 0: a hand-shake wait timed out)
    // compact Jacobian written by the fused prepare + gate kernel (VuPrepareArgs::fused): column u of a record is state column
    // acol[u]; the record holds na = 7 * (rows / (2 ncam)) + 1 columns with the record's row count as leading dimension. MODE 2 only.
    const int *acol;                  // [records][na_max] or null (dense H)
    int na_max, ncam;
    size_t h_stride;                  // doubles between the H records
    int v_stride;                     // doubles between the v records (= nr unless the launch serves one length class of a longer-strided batch)
    // Long tracks (49 .. 96 rows: 13 .. 21 stereo poses) as TWO sequential block updates of at most 48 rows each, which is the same
    // posterior (the blocks' measurement noises are independent: R = r^2 I, ekf.cpp:771) and keeps every update on the P-resident MODE 2
    // kernel: half 1 = the first h1 = 2 ceil(rows / 4) rows (the first camera's), half 2 = the rest (the second camera's) with its
    // innovation corrected by the first block's mean step, v2 - H2 dm1. Half 1 leaves dm1 in dm_out and skips the quaternion
    // normalisation; half 2 reads it (dm_in) and normalises. 0 = the whole record in one launch. Compact H, MODE 2 only.
    int half, nr_full;                // nr_full: rows of the whole record when neither nr_rec nor nr says so (uniform long tracks)
    double *dm_out; const double *dm_in;   // [batch][n]
    // half 1 only: the visit's gate statuses. A block 1 that meets a non-positive pivot applies nothing and turns its record's status
    // into CHI2 -- what every other mode reports for a broken factorisation --, so that block 2 (require_inlier) skips the record
    // instead of applying half an update on a stale dm_in (r03 advisor)
    int *gate_rw;
    // speculative frame loop over LONG records (spec 2 with half 1 / 2, r04): the record a filter applies is chosen by the scan of the
    // half-1 launch; `half_auto` makes the block split a per-record decision -- a record of at most 48 rows is applied whole by the
    // half-1 launch (normalisation, success count and cursor included) and skipped by the half-2 launch, a longer one takes both --,
    // and sel_io [batch] carries the chosen track of a long record (-1: none) from the half-1 launch to the half-2 launch, which
    // must not scan again: the first launch may have moved the cursor or turned the record's gate status into CHI2
    int half_auto; int *sel_io;
    // half_auto: the epochs of the frame's records, [n_tracks][batch] (VuPrepareArgs::epoch). A long record whose block 1 was applied and
    // whose block 2 then meets a non-positive pivot leaves (m, P) changed without a counted success: the filter's pending records are
    // marked "never prepared" (-1) so that the next pass prepares them against the state as it is now (r04 advisor)
    int *epoch;
    int batch;
    const int *rec_count, *rec_list;  // compaction list (VuPrepareArgs): workgroup i updates filter rec_list[i], i < *rec_count; the others exit
};

// spec 3 hand-shake between the workgroups of one filter (agent scope: they run on different CUs)
__device__ __forceinline__ void spec_publish(const UpdateArgs &a, int rec, int code)
{
    __hip_atomic_store(a.pub + rec, a.pass_id * 4 + code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// decision of record `rec` in this pass; bounded spin (0.2 s): a decision that never arrives is reported as -1
__device__ __forceinline__ int spec_wait(const UpdateArgs &a, int rec)
{
    for (int spin = 0; spin < (1 << 20); ++spin) {
        const int v = __hip_atomic_load(a.pub + rec, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 2) == a.pass_id) return v & 3;
        __builtin_amdgcn_s_sleep(8);
    }
    // never silently: the frame's result is then NOT the sequential loop's (a.err is read by hv_ekf_frame_error / the host-pointer entry points)
    if (a.err) atomicOr(a.err, 1);
    return -1;
}
// thread 0 of the workgroup of the LAST track, when that workgroup applies nothing: if no pending track of the filter was an
// inlier, nobody else moves the cursor -- every pending status is final (or the pass did not run: quota used up)
__device__ __forceinline__ void spec_close(const UpdateArgs &a, int b, int c0, bool pass_ran)
{
    bool any = false;
    for (int jj = c0; jj < a.n_tracks - 1 && !any; ++jj) any = spec_wait(a, jj * (int)gridDim.x + b) == 2;
    if (!any) a.cursor_out[b] = pass_ran ? a.n_tracks : c0;
}

constexpr int UPD_THREADS = 512;   // 8 waves = 2 per SIMD: 256 VGPRs each (whole column blocks of P stay in registers)

// MODE 0: T in the global workspace (tall matrix larger than LDS)
// MODE 1: T in LDS, H streamed from L2
// MODE 2: T and H in LDS (n <= 160, nr <= 48): K-split products, P read from HBM exactly once
// TI (MODE 2 only): 16-row tiles of H, nr <= 16 TI: a compile-time count keeps the MFMA loops free of
// branches (a uniform branch per tile made hipcc wait for each LDS operand right before its MFMA).
// AUTO: the half_auto form of the speculative loop over long records (its own kernel: the other instantiations keep their registers)
template <int MODE, int TI, bool AUTO = false>
__device__ __forceinline__ void ekf_update_body(const UpdateArgs &a, const int b_in /* filter: blockIdx.x, or an entry of a compaction list */)
{
    // (a list entry is a per-lane load: on the scalar unit the filter's base addresses are uniform and its gathers become scalar base +
    //  32-bit lane offset -- one VGPR and one VALU instruction per request instead of a 64-bit multiply-add pair)
    const int b = __builtin_amdgcn_readfirstlane(b_in);
    constexpr bool USE_LDS = MODE >= 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int e = b;                                             // record of this workgroup's H, v, active, chi2, status
    int sel = -1;
    if (a.spec == 1) {
        const int j = blockIdx.y;
        if (j < a.cursor[b] || a.success_counter[b] >= a.max_successful) return;
        e = j * (int)gridDim.x + b;
    } else if (AUTO && a.spec == 2 && a.half == 2) {
        // second block of the long record the half-1 launch chose for this filter (if any)
        sel = a.sel_io[b];
        if (sel < 0) return;
        e = sel * (int)gridDim.x + b;
        if (a.gate_in[e] != 0) {                           // block 1 met a non-positive pivot (status CHI2 now): nothing applied, the track is final
            if (threadIdx.x == 0) a.cursor[b] = sel + 1;
            return;
        }
    } else if (a.spec == 2) {
        // every thread runs the same short scan (uniform): the first pending track that passed its gate
        const int c0 = a.cursor[b];
        if (AUTO && threadIdx.x == 0) a.sel_io[b] = -1;      // (thread 0 also writes the choice below: program order)
        if (c0 >= a.n_tracks || a.success_counter[b] >= a.max_successful) return;
        for (int j = c0; j < a.n_tracks && sel < 0; ++j) {
            const int ee = j * (int)gridDim.x + b;
            if (a.active[ee] && a.gate_in[ee] == 0) sel = j;
        }
        if (sel < 0) {                                     // nothing applicable is left: every pending status is final
            __syncthreads();
            if (threadIdx.x == 0) a.cursor[b] = a.n_tracks;
            return;
        }
        e = sel * (int)gridDim.x + b;
    }
    int c0_spec = 0;
    if (a.spec == 3) {
        const int j = blockIdx.y;
        c0_spec = a.cursor[b];
        e = j * (int)gridDim.x + b;
        const bool pass_runs = c0_spec < a.n_tracks && a.success_counter[b] < a.max_successful;
        if (!pass_runs || j < c0_spec || !a.active[e]) {      // nothing to gate here: still publish, and close the pass if last
            if (threadIdx.x == 0) {
                spec_publish(a, e, 1);
                if (j == a.n_tracks - 1) spec_close(a, b, c0_spec, pass_runs);
            }
            return;
        }
    }
    // The record's flags and its shape are requested TOGETHER: as a sequence of tests each of them was
    // a dependent HBM / L2 round trip of its own -- 15 k cycles went by before the first request for H was issued (r03 phase stamps)
    const int t = threadIdx.x, lane = t & 63;
    const unsigned char act_v = a.active ? a.active[e] : (unsigned char)1;
    const int req_v = a.require_inlier ? a.require_inlier[b] : 0;
    const int nr_rec_v0 = a.nr_rec ? a.nr_rec[e] : a.nr_full;
    int shape_pin = nr_rec_v0;
    asm volatile("" : "+v"(shape_pin));                     // (keeps the three requests in front of the first test)
    const int nr_rec_v = shape_pin;
    if (!act_v) return;
    if (req_v != 0) return;
#ifdef HV_DEBUG_SKIP_UPDATE
    // experiment only (scripts/lanes_probe.py, never in the shipped library): what do the update workgroups cost a step? An applied
    // update counts as a success and leaves (m, P) as they are
    if (a.mode == 1 && a.spec == 0) { if (a.success_counter && t == 0) a.success_counter[b] += 1; return; }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // wave-uniform: tile indices and their addresses stay on the scalar unit
    constexpr int nwaves = UPD_THREADS / 64;
    const int n = a.n, l = a.l;
    int nr = a.nr, R = a.Rs, Rfull = a.R;                   // R is the column STRIDE of T below; Rfull rows are used
    int roff = 0, ld = a.nr;                               // first row of this launch's block inside the record; leading dimension of H
    int half = a.half;                                     // (half_auto: this record's own split, see UpdateArgs)
    if (a.nr_rec || a.half) {                              // ragged batch / block update: this record's own shape (uniform per workgroup)
        int nr_full = __builtin_amdgcn_readfirstlane(nr_rec_v);
        if (nr_full < 1 || nr_full > a.v_stride) return;    // no track (the prepare launch also cleared `active`)
        nr = nr_full; ld = nr_full;
        if constexpr (AUTO) {
            if (nr_full <= 48) {
                if (half == 2) return;
                half = 0;
            }
            if (half == 1 && t == 0) a.sel_io[b] = sel;
        }
        if (half) {
            if (MODE != 2) return;                         // (the host only asks the LDS-resident kernels for block updates)
            const int h1 = 2 * ((nr_full + 3) >> 2);
            roff = half == 2 ? h1 : 0;
            nr = half == 2 ? nr_full - h1 : h1;
        }
        if (nr < 1 || nr > a.nr) return;
        Rfull = nr + n + 1; R = Rfull;
        if (USE_LDS) while ((R & 31) != 15 && (R & 31) != 17) R++;
    }
    double *m = a.m + (size_t)b * n, *P = a.P + (size_t)b * n * n;
    const double *H = a.H + (size_t)e * a.h_stride;        // record stride: the launch's row count x columns; leading dimension: nr
    const double rd = a.rdiag ? a.rdiag[b] : a.rd0;
    // Tall matrix T (a.R rows x nr columns, column-major with stride R >= a.R: T(r, c) = T[c * R + r]; in LDS
    // the stride is padded to 15 or 17 mod 32 doubles, see ekf_launch_update):
    //   rows 0 .. nr-1   S = H P H' + R            -> L           (S = L L')
    //   row  nr          v'                        -> z' = (L^-1 v)'
    //   rows nr+1 ..     (H P)'  (n rows)          -> Y' = (L^-1 H P)'
    // One blocked Cholesky pass over T does the factorisation and both triangular solves. The gate
    // only needs rows 0 .. nr. All LDS comes from the dynamic region (keeps the base 16-byte aligned):
    // [T] W[256] col[320] red[16] flag. USE_LDS is a template parameter so that the common case compiles to
    // ds_read / ds_write (a run-time select would turn every access into a flat load).
    double *T = USE_LDS ? smem : a.ws + (size_t)b * a.Rs * a.nr;
    double *W = USE_LDS ? smem + (((size_t)R * nr + 1) & ~(size_t)1) : smem;   // inverse of the current diagonal block
    double *col = W + 256;                                                      // column broadcast buffer of the diagonal factor
    double *red = col + 544;                                                    // col[256 .. 527] is the dump area of factor_diag_block's branch-free stores
    int *s_stop = reinterpret_cast<int *>(red + nwaves);
    double *Hs = red + nwaves + 2;                // MODE 2: H zero-padded to (16 TI) x (16 lb), column-major, stride nrp
    constexpr int nrp = 16 * (TI > 0 ? TI : 1);
    const bool gate_only = a.mode == 0;
    const int rv = nr, ry = nr + 1;
    // mode 3 (MODE 2 kernels only): visualTrackOutlierCheck with R = rd0, then -- where it passes -- updateVisualTrack with
    // R = rd1 on the SAME H P: S (without R) and v are parked in the H staging area after phase B, the gate runs the
    // Cholesky on the measurement rows only, and an inlier restores S + rd1 I and v and runs the full factorisation.
    const bool two_r = MODE == 2 && a.mode == 3;
    int Rlim = (gate_only || two_r) ? nr + 1 : Rfull;

    PHASE_STAMP(0);
    const int kq = lane >> 4, cl = lane & 15;      // MFMA lane coordinates: k sub-step / output row group, column
    // MODE 2 ownership: wavefront w owns the 16-column block J = w of P with all of its 16-row K
    // blocks ("item 0"); wavefronts 0..3 also own one K half of block 8 or 9 ("item 1": J = 8 + (w & 1),
    // half w >> 1), which gives every SIMD 2.5 column blocks of matrix work at n = 160. The four B
    // operands of a K block ARE the 16 x 16 tile P(K, J) in the MFMA C layout (lane (kq, c) holds
    // rows kq + 4 s), so the tiles fetched for H P stay in registers and serve as the old values of
    // P -= Y'Y: P crosses HBM once per update. All of a wave's P loads are issued before anything
    // else; K blocks are consumed in arrival order, so the HBM stream (~10 B/clk/CU when every CU
    // pulls at once) overlaps the MFMAs.
    constexpr int NBK = 10, NBH = 5;                // K blocks per column block / per half (n <= 160)
    double pres0[NBK][4], pres1[NBH][4];
    const int tiles_j = (n + 15) >> 4, lb = (l + 15) >> 4;
    const int hs = (tiles_j + 1) >> 1;
    const int J1 = 8 + (wave & 1), kb0_1 = (wave & 2) ? hs : 0, kb1_1 = (wave & 2) ? tiles_j : hs;
    const bool have0 = MODE == 2 && wave < tiles_j, have1 = MODE == 2 && wave < 4 && J1 < tiles_j;
    unsigned kbmask = 0xFFFFFFFFu;                  // K blocks of H P with matrix work (compact H: those that hold a non-zero column)
    if constexpr (MODE == 2) {
        // ---- A: HP = H * P[0:l, :] accumulated into rows ry.. of T (the two K halves of blocks 8, 9
        // meet through ds_add_f64 on the zeroed tile: two addends commute, the sum is reproducible) ----
        auto fetch_blk = [&](auto &pv, int bi, int J, int kb, int kend) {
            if (kb < kend && (kb < lb || !gate_only)) {
                const int jc = min(J * 16 + cl, n - 1);
#pragma unroll
                for (int sx = 0; sx < 4; sx++)
                    pv[bi][sx] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) + (unsigned)(min(kb * 16 + 4 * sx + kq, n - 1) * n + jc) * 8u);
            }
        };
        // request order = arrival order: H first (every MFMA needs it), then the P tiles in the order
        // they are consumed. Only DEPTH K blocks are requested ahead of the MFMAs that use them: a
        // wave that queues its whole column block up front sits in the issue queue until most of it has
        // arrived, and the HBM stream no longer overlaps the matrix work.
        constexpr int HREG = (48 * 160 + UPD_THREADS - 1) / UPD_THREADS, DEPTH = 4;
        if (a.acol) {
            // compact H (fused prepare + gate): Hs is zeroed and the record's na x nr values are gathered BY COMPACT COLUMN and scattered
            // to their state columns -- the addresses of the gather do not depend on the column list, so Hc and the list travel
            // together: one HBM / L2 round trip, 8 values per thread at 11 stereo poses. (r03's first form built the inverse map
            // state column -> compact column in LDS first -- a dependent round trip and two barriers -- and walked all 48 x 160 cells
            // with 15 registers per thread.) kbmask: the 16-row K blocks of H P that hold a non-zero column of H at all.
            int *mask = reinterpret_cast<int *>(col);       // (the col scratch is free until the Cholesky)
            const int na = 7 * (ld / (2 * a.ncam)) + 1;     // (ld = rows of the whole record)
            const int *acol = a.acol + (size_t)e * a.na_max;
            const int total = na * nr;
            const unsigned inv_nr = (unsigned)((0x100000000ull + (unsigned)nr - 1) / (unsigned)nr);      // i / nr = umulhi(i, ceil(2^32 / nr)), i < 2^20
            constexpr int HC = 8;
            double hv[HC]; int hc[HC];
            auto request = [&](int base) {
#pragma unroll
                for (int u = 0; u < HC; u++) {
                    const int i = base + t + u * UPD_THREADS, uc = (int)__umulhi((unsigned)i, inv_nr), r = i - uc * nr;
                    const bool live = i < total;
                    hc[u] = live ? acol[uc] : -1;
                    hv[u] = live ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(H) + (unsigned)(uc * ld + roff + r) * 8u) : 0.0;
                }
            };
            auto scatter = [&](int base) {
#pragma unroll
                for (int u = 0; u < HC; u++) {
                    const int i = base + t + u * UPD_THREADS, uc = (int)__umulhi((unsigned)i, inv_nr), r = i - uc * nr;
                    if (hc[u] >= 0 && hc[u] < l) {
                        Hs[hc[u] * nrp + r] = hv[u];
                        if (r == 0) atomicOr(mask, 1 << (hc[u] >> 4));
                    }
                }
            };
            request(0);
            // The P tiles are requested BEHIND the gather, the block that is used last FIRST: the kernel sits at 255 VGPRs and the
            // register allocator spills one value of that block right behind its load, i.e. waits for it -- and loads return in order.
            // (r03's first form requested the tiles in front of the staging, block 0 first: the spill waited for all 16 of them,
            // 14 k cycles before the first request for H.)
#pragma unroll
            for (int bi = DEPTH - 1; bi >= 0; bi--) if (have0) fetch_blk(pres0, bi, wave, bi, tiles_j);     // (LAST block first: see below)
            PHASE_STAMP(8);
            for (int i = t; i < nrp * 16 * lb; i += UPD_THREADS) Hs[i] = 0.0;
            for (int i = t; i < R * nr; i += UPD_THREADS) T[i] = 0.0;
            if (t == 0) mask[0] = 0;
            __syncthreads();
            scatter(0);
            for (int base = HC * UPD_THREADS; base < total; base += HC * UPD_THREADS) { request(base); scatter(base); }     // (long mono tracks)
            __syncthreads();
            kbmask = (unsigned)mask[0];
        } else {
            double hreg[HREG];
#pragma unroll
            for (int u = 0; u < HREG; u++) {
                const int i = t + u * UPD_THREADS, k = i / nrp, r = i - k * nrp;
                hreg[u] = (i < nrp * 16 * lb && k < l && r < nr) ? H[(size_t)k * ld + roff + r] : 0.0;
            }
#pragma unroll
            for (int bi = 0; bi < DEPTH; bi++) if (have0) fetch_blk(pres0, bi, wave, bi, tiles_j);
            PHASE_STAMP(8);
            for (int i = t; i < R * nr; i += UPD_THREADS) T[i] = 0.0;
#pragma unroll
            for (int u = 0; u < HREG; u++) {
                const int i = t + u * UPD_THREADS;
                if (i < nrp * 16 * lb) Hs[i] = hreg[u];
            }
        }
        __syncthreads();
        PHASE_STAMP(9);
        const double *hbase = Hs + (size_t)kq * nrp + cl;
        // H operands of a K block: 4 k-steps x TI row tiles, double-buffered across blocks (the reads of
        // block b+1 are issued before the MFMAs of block b: a wavefront that only prefetches one or two
        // ds_reads ahead issues an MFMA every ~100 cycles instead of every 64)
        // (in HALF blocks of two k-steps: 2 x 2 x TI operand registers instead of 2 x 4 x TI -- the kernel sits at 255 VGPRs and the
        //  allocator took its last pair from the P prefetch, spilling a value right behind its load: a full stop in front of the staging)
        auto load_h = [&](int kb, int hf, double (&av)[2][TI]) {
#pragma unroll
            for (int sx = 0; sx < 2; sx++)
#pragma unroll
                for (int mt = 0; mt < TI; mt++) av[sx][mt] = hbase[(size_t)(kb * 16 + 4 * (2 * hf + sx)) * nrp + 16 * mt];
        };
        auto mfma_half = [&](auto &pv, int bi, int hf, const double (&av)[2][TI], double4v (&acc)[TI]) {
#pragma unroll
            for (int sx = 0; sx < 2; sx++)
#pragma unroll
                for (int mt = 0; mt < TI; mt++)
                    acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[sx][mt], pv[bi][2 * hf + sx], acc[mt], 0, 0, 0);
        };
        auto flush = [&](int J, double4v (&acc)[TI]) {
#pragma unroll
            for (int mt = 0; mt < TI; mt++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = 16 * mt + kq + 4 * q, j = J * 16 + cl;
                    if (i < nr && j < n) unsafeAtomicAdd(&T[(size_t)i * R + ry + j], acc[mt][q]);
                }
            }
        };
        {
            double4v acc[TI];
#pragma unroll
            for (int mt = 0; mt < TI; mt++) acc[mt] = double4v{0.0, 0.0, 0.0, 0.0};
            const int nv = have0 ? min(tiles_j, lb) : 0;                   // K blocks with matrix work: 0 .. nv-1
            double av[2][2][TI];
            if (nv > 0) load_h(0, 0, av[0]);
#pragma unroll
            for (int bi = 0; bi < NBK; bi++) {
                if (bi + DEPTH < NBK) { if (have0) fetch_blk(pres0, bi + DEPTH, wave, bi + DEPTH, tiles_j); }
                else if (bi + DEPTH - NBK < NBH) { if (have1) fetch_blk(pres1, bi + DEPTH - NBK, J1, kb0_1 + bi + DEPTH - NBK, kb1_1); }
                if (bi < nv) {
                    const bool work = (kbmask >> bi) & 1;
                    load_h(bi, 1, av[1]);
                    if (work) mfma_half(pres0, bi, 0, av[0], acc);
                    if (bi + 1 < nv) load_h(bi + 1, 0, av[0]);
                    if (work) mfma_half(pres0, bi, 1, av[1], acc);
                }
            }
            if (have0) flush(wave, acc);
        }
        PHASE_STAMP(10);
        {
            double4v acc[TI];
#pragma unroll
            for (int mt = 0; mt < TI; mt++) acc[mt] = double4v{0.0, 0.0, 0.0, 0.0};
            const int nv = have1 ? max(min(kb1_1, lb) - kb0_1, 0) : 0;     // blocks kb0_1 .. kb0_1 + nv - 1
            double av[2][2][TI];
            if (nv > 0) load_h(kb0_1, 0, av[0]);
#pragma unroll
            for (int bi = 0; bi < NBH; bi++) {
                if (bi + DEPTH < NBH) { if (have1) fetch_blk(pres1, bi + DEPTH, J1, kb0_1 + bi + DEPTH, kb1_1); }
                if (bi < nv) {
                    const bool work = (kbmask >> (kb0_1 + bi)) & 1;
                    load_h(kb0_1 + bi, 1, av[1]);
                    if (work) mfma_half(pres1, bi, 0, av[0], acc);
                    if (bi + 1 < nv) load_h(kb0_1 + bi + 1, 0, av[0]);
                    if (work) mfma_half(pres1, bi, 1, av[1], acc);
                }
            }
            if (have1) flush(J1, acc);
        }
        PHASE_STAMP(11);
    } else {
        // ---- A: HP = H * P[0:l, :], stored transposed as rows ry .. ry+n-1 of T; residual row ----
        const int tiles_i = (nr + 15) / 16;
        for (int tile = wave; tile < tiles_i * tiles_j; tile += nwaves) {
            const int i0 = (tile % tiles_i) * 16, j0 = (tile / tiles_i) * 16;
            // A(i, k) = H(i0+i, k) = H[k*nr + i0+i];  B(k, j) = P(k, j0+j), read as P(j0+j, k) = P[k*n + j0+j]:
            // P is symmetric to rounding (predict builds P01/P10 separately, every other step keeps or
            // restores symmetry) and the transposed element is unit-stride across the 16 lanes of a k-row
            const double4v acc = mfma_tile<8>(H + i0, 1, nr, nr - i0, P + j0, n, 1, n - j0, l);
            const int j = j0 + cl;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + kq + 4 * q;
                if (i < nr && j < n) T[(size_t)i * R + ry + j] = acc[q];
            }
        }
    }
    for (int i = t; i < nr; i += UPD_THREADS) {
        double r = a.v[(size_t)e * a.v_stride + roff + i];
        if (a.generic) { double s = 0; for (int k = 0; k < l; k++) s += H[(size_t)k * nr + i] * m[k]; r -= s; }
        if constexpr (MODE == 2) {
            if (a.dm_in) {                                 // second block of a long track: v2 - H2 dm1 (Hs: the staged, zero-padded H of this block)
                const double *dm = a.dm_in + (size_t)b * n;
                double s = 0;
                for (int k = 0; k < l; k++) s += Hs[(size_t)k * nrp + i] * dm[k];
                r -= s;
            }
        }
        T[(size_t)i * R + rv] = r;
    }
    __syncthreads();

    PHASE_STAMP(1);
    // ---- B: S = HP[:, 0:l] * H' + R (lower triangle), rows 0 .. nr-1 of T ----
    {
        const int tb = (nr + 15) / 16;
        if constexpr (MODE == 2) {
            // (tile, K half) items, the halves combined with ds_add_f64 into the zeroed S block
            const int khalf = ((l / 2) + 15) & ~15;
            for (int it = wave; it < tb * (tb + 1); it += nwaves) {             // lower tiles only, two halves each
                const int hh = it & 1;
                int ib = it >> 1, cb = 0;
                while (ib >= tb - cb) { ib -= tb - cb; cb++; }                   // column cb holds tiles ib = cb .. tb-1
                ib += cb;
                const int i0 = ib * 16, c0 = cb * 16;
                const int kbeg = hh ? min(khalf, l) : 0, klen = hh ? l - kbeg : min(khalf, l);
                if (klen <= 0) continue;
                // A(i, k) = HP(i0+i, k) = T[(i0+i)*R + ry + k];  B(k, c) = H(c0+c, k) = Hs[k*nrp + c0+c]
                // (Hs is zero beyond k = l and c = nr; rows i >= nr are clamped and never stored)
                const double *pa = T + (size_t)min(i0 + cl, nr - 1) * R + ry + kbeg + kq;
                const double *pb = Hs + (size_t)(kbeg + kq) * nrp + c0 + cl;
                double4v acc = {0.0, 0.0, 0.0, 0.0};
                for (int k0 = 0; k0 < klen; k0 += 32) {          // 8 k-steps per trip, all 16 operands requested first
                    double av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool live = k0 + 4 * u < klen;             // a partial last step meets the zero padding of Hs
                        const int kk = live ? k0 + 4 * u : 0;
                        av[u] = pa[kk]; bv[u] = pb[(size_t)kk * nrp];
                        if (!live) bv[u] = 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
                }
                const int c = c0 + cl;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + kq + 4 * q;
                    if (i < nr && c < nr) unsafeAtomicAdd(&T[(size_t)c * R + i], acc[q] + ((i == c && hh == 0) ? rd : 0.0));
                }
            }
        } else {
            for (int tile = wave; tile < tb * tb; tile += nwaves) {
                const int ib = tile % tb, cb = tile / tb;
                if (ib < cb) continue;
                const int i0 = ib * 16, c0 = cb * 16;
                // A(i, k) = HP(i0+i, k) = T[(i0+i)*R + ry + k];  B(k, c) = H(c0+c, k) = H[k*nr + c0+c]
                const double4v acc = mfma_tile<8>(T + (size_t)i0 * R + ry, R, 1, nr - i0, H + c0, nr, 1, nr - c0, l);
                const int c = c0 + cl;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + kq + 4 * q;
                    if (i < nr && c < nr) T[(size_t)c * R + i] = acc[q] + (i == c ? rd : 0.0);
                }
            }
        }
    }
    __syncthreads();
    if constexpr (MODE == 2) {
        if (two_r) {                                          // Hs is dead after phase B: (nr + 1) x nr doubles fit (nrp x 16 lb >= that)
            // (+ 256: slack after the header area)
            for (int e = t; e < (nr + 1) * nr; e += UPD_THREADS) { const int c = e / (nr + 1), i = e - c * (nr + 1); Hs[256 + e] = T[(size_t)c * R + i]; }
            __syncthreads();
        }
    }

    PHASE_STAMP(2);
    for (int pass = 0; pass < (two_r ? 2 : 1); ++pass) {
    // ---- C: blocked left-looking Cholesky of the tall matrix, 16 columns per block, 3 barriers per
    // block (a column-at-a-time version needs one barrier + one LDS round trip + one rsqrt per COLUMN
    // on the critical path of all 16 waves: 1.3-1.7 k cycles per column measured). Per block j:
    //   U  every 16-row tile of block column j -= T(tile, 0:j0) * T(j0:j0+16, 0:j0)'         (MFMA)
    //   D  wave 0 factors the diagonal block in registers and inverts the factor   (factor_diag_block)
    //   P  every tile below the diagonal block *= Linv'                                       (MFMA)
    // Multiplying by the explicit inverse of a 16 x 16 diagonal block instead of substituting is the
    // usual blocked-TRSM trade: the error grows with cond(L_jj), not cond(L).
    for (int j0 = 0; j0 < nr; j0 += 16) {
        const int w = min(16, nr - j0);
        const int ntile = (Rlim - j0 + 15) / 16;
        const int i_l = lane >> 4, c_l = lane & 15;
        if (j0 > 0) {
            for (int tile = wave; tile < ntile; tile += nwaves) {
                const int i0 = j0 + 16 * tile, mi = Rlim - i0;
                // A(i, k) = T(i0+i, k) = T[k*R + i0+i];  B(k, c) = T(j0+c, k) = T[k*R + j0+c]
                const double4v acc = mfma_tile(T + i0, 1, R, mi, T + j0, R, 1, w, j0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i_l + 4 * q;
                    if (i < mi && c_l < w) T[(size_t)(j0 + c_l) * R + i0 + i] -= acc[q];
                }
            }
            __syncthreads();
        }
        if (j0 == 0) PHASE_STAMP(6);
        if (wave == 0) { if (w <= 8) factor_diag_block<8>(T, W, col, R, j0, w, lane); else factor_diag_block<16>(T, W, col, R, j0, w, lane); }
        __syncthreads();
        if (j0 == 0) PHASE_STAMP(7);
        for (int i0 = j0 + w + 16 * wave; i0 < Rlim; i0 += 16 * nwaves) {        // rows below the w x w diagonal block
            const int mi = Rlim - i0;
            // A(i, k) = T(i0+i, j0+k) = T[(j0+k)*R + i0+i];  B(k, c) = Linv(c, k) = W[k*16 + c]
            const double4v acc = mfma_tile(T + (size_t)j0 * R + i0, 1, R, mi, W, 16, 1, 16, w);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i_l + 4 * q;
                if (i < mi && c_l < w) T[(size_t)(j0 + c_l) * R + i0 + i] = acc[q];
            }
        }
        __syncthreads();
    }

    PHASE_STAMP(3);
    // ---- D: chi2 = noise_scale * z'z ----
    {
        double s = 0;
        for (int c = t; c < nr; c += UPD_THREADS) { const double z = T[(size_t)c * R + rv]; s += z * z; }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (t == 0) {
            double tot = 0; for (int w2 = 0; w2 < nwaves; w2++) tot += red[w2];
            tot *= a.noise_scale;
            // A non-positive pivot (S not positive definite: r = 0 with a rank-deficient H) leaves NaN / inf in z: such a filter is
            // reported as CHI2 and left untouched in EVERY mode, where the reference's pivoted LDLT would carry on (r01 advisor)
            const bool broken = !(tot < 1e300);
            const int outlier = broken || ((nr < HV_CHI2INV95_N) ? (tot > d_chi2inv95[nr]) : 0);
            if (pass == 0) {
                if (a.chi2) a.chi2[e] = tot;
                if (a.status) a.status[e] = outlier ? 3 /*CHI2*/ : 0 /*INLIER*/;
            }
            *s_stop = (a.mode == 0) || broken || ((a.mode == 2 || (two_r && pass == 0)) && outlier);
            if (broken && a.gate_rw) a.gate_rw[e] = 3 /*CHI2*/;
            if (AUTO && broken && a.spec == 2 && half != 1) a.cursor[b] = sel + 1;     // (whole short record or block 2: not applied, the track is final)
            if (AUTO && broken && a.spec == 2 && half == 2 && a.epoch)                 // block 1 of this record HAS been applied
                for (int jj = sel + 1; jj < a.n_tracks; ++jj) a.epoch[jj * (int)gridDim.x + b] = -1;
            if (a.spec == 3 && pass == 0) {
                const int j = blockIdx.y;
                spec_publish(a, e, outlier ? 1 : 2);
                if (!outlier) {                               // first inlier in visit order applies; the others are re-examined next pass
                    bool lose = false;
                    for (int jj = c0_spec; jj < j && !lose; ++jj) lose = spec_wait(a, jj * (int)gridDim.x + b) != 1;
                    if (lose) *s_stop = 1;
                }
                if (*s_stop && j == a.n_tracks - 1 && outlier) spec_close(a, b, c0_spec, true);
            }
            // the applying workgroup met a non-positive pivot in the second factorisation: nothing is applied, the track is final
            if (a.spec == 3 && pass == 1 && *s_stop) a.cursor_out[b] = (int)blockIdx.y + 1;
        }
        __syncthreads();
        if (*s_stop) return;
    }
    if constexpr (MODE == 2) {
        if (two_r && pass == 0) {                             // inlier: S + rd1 I and v back, then the full factorisation
            const double shift = a.rd1 - rd;
            for (int e = t; e < (nr + 1) * nr; e += UPD_THREADS) {
                const int c = e / (nr + 1), i = e - c * (nr + 1);
                T[(size_t)c * R + i] = Hs[256 + e] + (i == c ? shift : 0.0);
            }
            Rlim = Rfull;
            __syncthreads();
        }
    }
    }   // pass

    PHASE_STAMP(4);
    // ---- E: m += Y' z ----
    for (int j = t; j < n; j += UPD_THREADS) {
        double s = 0;
        for (int c = 0; c < nr; c++) s += T[(size_t)c * R + ry + j] * T[(size_t)c * R + rv];
        m[j] += s;
        if (a.dm_out && !(AUTO && half == 0)) a.dm_out[(size_t)b * n + j] = s;
    }
    // ---- F: P -= Y' Y ----
    if constexpr (MODE == 2) {
        // tiles P(K, J) of the wave's items; old values are the registers filled in phase A
        auto down_item = [&](auto &pv, int nblk, int J, int kb0, int kb1) {
            const int jc = min(J * 16 + cl, n - 1);
            // B(c, j) = Y(c, J*16 + j) = T[c*R + ry + J*16 + j], c = 4 s + kq: shared by all tiles of the item
            constexpr int KS = 4 * TI;              // k-steps over the nr <= 16 TI rows of Y (zero beyond nr)
            double yj[KS];
#pragma unroll
            for (int sx = 0; sx < KS; sx++) {
                const int c = 4 * sx + kq;
                const double y = T[(size_t)min(c, nr - 1) * R + ry + jc];
                yj[sx] = c < nr ? y : 0.0;
            }
#pragma unroll
            for (int bi = 0; bi < nblk; bi++) {
                const int kb = kb0 + bi;
                if (kb < kb1) {
                    // A(i, c) = Y(c, kb*16 + i) = T[c*R + ry + kb*16 + i]; rows c >= nr meet yj = 0
                    const int ic = min(kb * 16 + cl, n - 1);
                    double av[KS];
#pragma unroll
                    for (int sx = 0; sx < KS; sx++) av[sx] = T[(size_t)min(4 * sx + kq, nr - 1) * R + ry + ic];
                    double4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int sx = 0; sx < KS; sx++) {
                        // only the last tile of rows can be (partly) empty: skip whole k-steps beyond nr
                        if (sx < KS - 4 || 4 * sx < nr) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[sx], yj[sx], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = kb * 16 + kq + 4 * q, j = J * 16 + cl;
                        if (i < n && j < n) P[(size_t)i * n + j] = pv[bi][q] - acc[q];
                    }
                }
            }
        };
        if (have0) down_item(pres0, NBK, wave, 0, tiles_j);
        if (have1) down_item(pres1, NBH, J1, kb0_1, kb1_1);
    } else {
        // Y'Y is symmetric: entry (i, j) of a tile is applied to element (j, i), so the 16 lanes of a
        // row group touch 16 consecutive doubles. The old values of the NEXT tile are requested before
        // the MFMA loop of the current one: the ~1 us HBM round trip of P hides behind the matrix work
        // instead of stalling every tile.
        const int tb = (n + 15) / 16, ntiles = tb * tb;
        const int cj = lane & 15, ri = lane >> 4;
        auto p_addr = [&](int tile, int q) -> double * {
            const int i = min((tile % tb) * 16 + ri + 4 * q, n - 1), j = min((tile / tb) * 16 + cj, n - 1);
            return P + (size_t)i * n + j;
        };
        double cur[4], nxt[4] = {0.0, 0.0, 0.0, 0.0};
        if (wave < ntiles) {
#pragma unroll
            for (int q = 0; q < 4; q++) cur[q] = *p_addr(wave, q);
        }
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const int i0 = (tile % tb) * 16, j0 = (tile / tb) * 16;
            const int tn = min(tile + nwaves, ntiles - 1);
#pragma unroll
            for (int q = 0; q < 4; q++) nxt[q] = *p_addr(tn, q);
            // A(i, c) = Y(c, i0+i) = T[c*R + ry + i0+i];  B(c, j) = Y(c, j0+j) = T[c*R + ry + j0+j]
            const double4v acc = mfma_tile(T + ry + i0, 1, R, n - i0, T + ry + j0, R, 1, n - j0, nr);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + ri + 4 * q, j = j0 + cj;
                if (i < n && j < n) P[(size_t)i * n + j] = cur[q] - acc[q];
                cur[q] = nxt[q];
            }
        }
    }
    __syncthreads();
    PHASE_STAMP(5);
    // ---- G: quaternion normalisation (updateCommon ekf.cpp:29-31 / normalizeQuaternions 1024-1032) ----
    const bool first_block = AUTO ? half == 1 : a.normalize_all < 0;                              // of a long track: no normalisation yet
    const int nq = first_block ? 0 : (a.normalize_all != 0) ? 1 + (n - a.map_dim - CAM) / POSE : 1;      // map points behind the trail are not poses
    if (t < nq) normalize4(m + (t == 0 ? ORI : CAM + POSE * (t - 1) + 3));
    if (a.success_counter && t == 0 && !(AUTO && half == 1)) a.success_counter[b] += 1;
    if (a.spec == 2 && t == 0 && !(AUTO && half == 1)) a.cursor[b] = sel + 1;      // the tracks up to the applied one are final, the rest is re-examined
    if (a.spec == 3 && t == 0) a.cursor_out[b] = (int)blockIdx.y + 1;
}

template <int MODE, int TI>
__global__ __launch_bounds__(UPD_THREADS) void ekf_update_kernel(UpdateArgs a)
{
    // (r03 also tried one workgroup per CU pulling filters from a device queue: inside a loop the body lost its register allocation --
    // ~480 spilled VGPRs, 2.4x the time per filter -- and keeping P in the registers across the two blocks of a long track in ONE launch
    // -- 250 - 450 spilled VGPRs --, so masked launches use the compaction list and long tracks take two launches)
    int b = blockIdx.x;
    if (a.rec_list) { if ((int)blockIdx.x >= *a.rec_count) return; b = a.rec_list[blockIdx.x]; }
    ekf_update_body<MODE, TI>(a, b);
}

// the update launches of the speculative loop over long records (UpdateArgs::half_auto), MODE 2, three row tiles
__global__ __launch_bounds__(UPD_THREADS) void ekf_update_spec_long_kernel(UpdateArgs a)
{
    ekf_update_body<2, 3, true>(a, blockIdx.x);
}

// Two masked update launches in ONE grid (ragged visits: the inliers of the short class and the first block of the long class's
// inliers are different filters, ~200 + ~50 of 1024 at the reference's track mix -- one workgroup each on a CU of its own, so two
// launches were two latency chains back to back on a mostly idle chip): the first workgroups serve a1's list, the next ones a0's.
// The argument block is chosen by a uniform select; the body is instantiated once.
// Every workgroup needs a CU of its own, so a grid of more than num_cus live workgroups takes a second round. The long class follows
// with a second launch anyway (its second block update): `part` 0 = all of a1 + the entries of a0's list that fit beside them on the
// chip (cap = num_cus - *a1.rec_count), `part` 1 = all of a1 + the REST of a0's list -- the visit loop issues (a0 = short class, a1 =
// long block 1, part 0) and then (a0 = short class, a1 = long block 2, part 1); `part` < 0: all of both lists.
struct UpdatePair { UpdateArgs a[2]; };
template <int MODE, int TI>
__global__ __launch_bounds__(UPD_THREADS) void ekf_update_dual_kernel(UpdatePair p, int part, int num_cus)
{
    const int n0 = *p.a[0].rec_count, n1 = *p.a[1].rec_count, i = (int)blockIdx.x;
    const int cap = min(n0, max(num_cus - n1, 0));
    const int lo = part == 1 ? cap : 0, hi = part == 0 ? cap : n0;          // a[0]'s share of this launch
    const bool second = i < n1;
    // (ONE block is read, through a run-time index into the kernel-argument segment: with two by-value blocks and a select per field
    //  both sat in SGPRs -- 250 of them spilled into VGPR lanes, in a kernel that has no VGPR to spare)
    const UpdateArgs &a = p.a[second ? 1 : 0];
    const int j = second ? i : i - n1 + lo;
    if (!second && j >= hi) return;
    ekf_update_body<MODE, TI>(a, a.rec_list[j]);
}

// (r05, measured and dropped: the same grid with a long record's workgroup running block 2 right behind block 1 -- P through L2 in
// between -- and a second launch that only serves the short records that found no CU: one lane 10.04 / 10.00 -> 10.13 / 10.10 ms per
// step, four lanes 29.44 / 29.27 -> 29.40 / 29.23: the visit's length is the long records' block-1 -> block-2 chain either way, and two
// instances of the body cost 12 more spilled VGPRs. profiles/r05/long_blocks_one_launch_ab.txt)

// ---------------------------------------------------------------------------------------------
// chi2 gate for MANY filters at once (throughput launches): S = H P H' + R without ever holding H P.
//
// ekf_update_kernel keeps P in registers (8 waves x 256 VGPRs) and the tall matrix in 134 KB of LDS so that an UPDATE reads P
// once -- one filter per CU at a time, its phases (products, S, a 40-pivot Cholesky) strictly one after the other: a gate launch over
// 1024 filters is 4 rounds of a 33 us latency chain at 32 % MFMA busy. A gate needs none of that state. Here a wavefront
// owns 16-column blocks J of P and chains two MFMA products per block through its accumulators:
//     G_J = P(J, :) H'          (16 x nr; A = tiles of P streamed from HBM, B = H' from LDS)
//     S  += H(:, J) G_J         (nr x nr; A = H from LDS, B = G_J -- the accumulator tile of the first product IS the B operand
//                                layout of the second: lane (kq, c) holds rows 4 v + kq, exactly the k order of a k-step)
// so H P never exists, LDS holds H (61 KB) + S (15 KB) and two workgroups of 4 waves share a CU: one filter's Cholesky runs under
// the other's MFMAs. P(J, k) is read as stored (no symmetry assumption). The factorisation / chi2 part is the blocked Cholesky of
// ekf_update_kernel restricted to the measurement rows.
// ---------------------------------------------------------------------------------------------
struct GateArgs {
    int n, nr, l, Rs;                 // Rs: column stride of the (nr + 1) x nr matrix [S; v'] in LDS
    const double *P;                  // [batch][n][n]
    const double *H, *v;              // [batch] records: nr x l column-major, nr
    double rd, noise_scale;
    double *chi2; int *status;
    const unsigned char *active;
    const int *success_counter; int max_successful;      // optional: filters whose quota is used up are skipped
};

constexpr int GATE_THREADS = 256;

template <int TI>
__global__ __launch_bounds__(GATE_THREADS, 2) void ekf_gate_stream_kernel(GateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    if (a.success_counter && a.success_counter[b] >= a.max_successful) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int nwaves = GATE_THREADS / 64, nrp = 16 * TI;
    const int n = a.n, nr = a.nr, l = a.l, R = a.Rs;
    const double *P = a.P + (size_t)b * n * n, *H = a.H + (size_t)b * nr * l;
    const int lb = (l + 15) >> 4;
    // LDS: Hs [16 lb][nrp] (k-major, zero padded) | T [(nr + 1) x nr, stride R] ; W / col / red live in Hs once the products are done
    double *Hs = smem, *T = smem + (size_t)nrp * 16 * lb;
    const int kq = lane >> 4, cl = lane & 15;
    // ---- stage H (zero padded) and zero T ----
    for (int i = t; i < nrp * 16 * lb; i += GATE_THREADS) {
        const int k = i / nrp, r = i - k * nrp;
        Hs[i] = (k < l && r < nr) ? H[(size_t)k * nr + r] : 0.0;
    }
    for (int i = t; i < R * nr; i += GATE_THREADS) T[i] = 0.0;
    __syncthreads();
    // ---- products: this wave's column blocks J = wave, wave + 4, .. ----
    double4v accS[TI * (TI + 1) / 2];
#pragma unroll
    for (int u = 0; u < TI * (TI + 1) / 2; u++) accS[u] = double4v{0.0, 0.0, 0.0, 0.0};
    constexpr int DEPTH = 4;                                    // K blocks of P requested ahead of their MFMAs
    for (int J = wave; J < lb; J += nwaves) {
        const int jrow = min(J * 16 + cl, n - 1);
        auto fetch = [&](double (&pv)[4], int kb) {
#pragma unroll
            for (int sx = 0; sx < 4; sx++) pv[sx] = P[(size_t)min(kb * 16 + 4 * sx + kq, n - 1) * n + jrow];   // A(i = cl, k): P(J16 + cl, k)
        };
        double pv[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) if (d < lb) fetch(pv[d], d);
        double4v accG[TI];
#pragma unroll
        for (int ct = 0; ct < TI; ct++) accG[ct] = double4v{0.0, 0.0, 0.0, 0.0};
        for (int kb0 = 0; kb0 < lb; kb0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const int kb = kb0 + d;
                if (kb < lb) {
                    double hb[4][TI];
#pragma unroll
                    for (int sx = 0; sx < 4; sx++)
#pragma unroll
                        for (int ct = 0; ct < TI; ct++) hb[sx][ct] = Hs[(size_t)(kb * 16 + 4 * sx + kq) * nrp + 16 * ct + cl];   // B(k, c) = H(c, k)
#pragma unroll
                    for (int sx = 0; sx < 4; sx++)
#pragma unroll
                        for (int ct = 0; ct < TI; ct++)
                            accG[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(pv[d][sx], hb[sx][ct], accG[ct], 0, 0, 0);
                    if (kb + DEPTH < lb) fetch(pv[d], kb + DEPTH);
                }
            }
        }
        // S(rt, ct) += H(16 rt + i, J16 + k) G_J(k, 16 ct + c), lower tiles; k-step v: A = H(.., J16 + 4 v + kq), B = accG[ct][v]
        if (J * 16 < n) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                double ha[TI];
#pragma unroll
                for (int rt = 0; rt < TI; rt++) ha[rt] = Hs[(size_t)(J * 16 + 4 * v + kq) * nrp + 16 * rt + cl];                 // A(i = cl, k)
                int u = 0;
#pragma unroll
                for (int ct = 0; ct < TI; ct++)
#pragma unroll
                    for (int rt = ct; rt < TI; rt++, u++)
                        accS[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ha[rt], accG[ct][v], accS[u], 0, 0, 0);
            }
        }
    }
    // ---- the waves' partial S meet in LDS in WAVE ORDER (r05: one wave per round, a barrier between the rounds -- with ds_add_f64 in
    // arrival order, r01 .. r04, two runs could differ by ~1 ulp of S) ----
    for (int w_turn = 0; w_turn < nwaves; ++w_turn) {
        if (wave == w_turn) {
            int u = 0;
#pragma unroll
            for (int ct = 0; ct < TI; ct++)
#pragma unroll
                for (int rt = ct; rt < TI; rt++, u++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = 16 * rt + kq + 4 * q, c = 16 * ct + cl;
                        if (i < nr && c < nr && i >= c) T[(size_t)c * R + i] += accS[u][q];
                    }
        }
        __syncthreads();
    }
    for (int i = t; i < nr; i += GATE_THREADS) { T[(size_t)i * R + i] += a.rd; T[(size_t)i * R + nr] = a.v[(size_t)b * nr + i]; }
    double *W = Hs, *col = Hs + 256, *red = Hs + 256 + 544;    // H is dead from here on
    __syncthreads();
    // ---- blocked Cholesky of [S; v'] (see ekf_update_kernel phase C), rows 0 .. nr ----
    const int Rlim = nr + 1;
    for (int j0 = 0; j0 < nr; j0 += 16) {
        const int w = min(16, nr - j0);
        const int ntile = (Rlim - j0 + 15) / 16;
        if (j0 > 0) {
            for (int tile = wave; tile < ntile; tile += nwaves) {
                const int i0 = j0 + 16 * tile, mi = Rlim - i0;
                const double4v acc = mfma_tile(T + i0, 1, R, mi, T + j0, R, 1, w, j0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = kq + 4 * q;
                    if (i < mi && cl < w) T[(size_t)(j0 + cl) * R + i0 + i] -= acc[q];
                }
            }
            __syncthreads();
        }
        if (wave == 0) { if (w <= 8) factor_diag_block<8>(T, W, col, R, j0, w, lane); else factor_diag_block<16>(T, W, col, R, j0, w, lane); }
        __syncthreads();
        for (int i0 = j0 + w + 16 * wave; i0 < Rlim; i0 += 16 * nwaves) {
            const int mi = Rlim - i0;
            const double4v acc = mfma_tile(T + (size_t)j0 * R + i0, 1, R, mi, W, 16, 1, 16, w);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = kq + 4 * q;
                if (i < mi && cl < w) T[(size_t)(j0 + cl) * R + i0 + i] = acc[q];
            }
        }
        __syncthreads();
    }
    // ---- chi2 = noise_scale * z'z ----
    double sz = 0;
    for (int c = t; c < nr; c += GATE_THREADS) { const double z = T[(size_t)c * R + nr]; sz += z * z; }
    for (int o = 32; o > 0; o >>= 1) sz += __shfl_down(sz, o);
    if (lane == 0) red[wave] = sz;
    __syncthreads();
    if (t == 0) {
        double tot = 0; for (int w2 = 0; w2 < nwaves; w2++) tot += red[w2];
        tot *= a.noise_scale;
        if (a.chi2) a.chi2[b] = tot;
        if (a.status) a.status[b] = tot > d_chi2inv95[nr] ? 3 /*CHI2*/ : 0 /*INLIER*/;
    }
}

// ---------------------------------------------------------------------------------------------
// Column-sparse chi2 gate as its own launch (many filters): visualTrackOutlierCheck on the compact Jacobians the prepare launch left
// in HBM (VuPrepareArgs::fused == 2). One 4-wave workgroup per filter -- four waves sit one per SIMD, so the (J, ct) items of
// sparse_gate spread evenly over the matrix pipes --, LDS = Hc (28 KB at 10 stereo poses) + [S; v'] (15 KB): three workgroups per CU,
// whose single-wave Cholesky chains (40 dependent pivots, ~10 us) run under each other's products. The same chain inside the
// prepare kernel (two 80 KB workgroups per CU, both latency chains themselves) cost 60 us per 1024 filters (r03 measurements).
// ---------------------------------------------------------------------------------------------
struct SparseGateArgs {
    int n, nr, ncam, na_max;          // state dimension; rows of the longest record (= record stride of v); cameras; columns per Hc record
    const double *P;                  // [batch][n][n]
    const double *Hc, *v;             // [batch][nr * na_max] compact Jacobians (leading dimension: the record's rows), [batch][nr]
    const int *acol;                  // [batch][na_max]
    const int *nr_rec;                // ragged batches: rows of every record, or null
    const unsigned char *active;      // [batch]: 1 where triangulation and prepareVisualUpdate passed (written by the prepare launch)
    double rd, noise_scale;
    double *chi2; int *status;        // chi2 optional
    int hs_doubles;                   // LDS carve: doubles reserved for the staged Hc (>= 816 + 4: it is the Cholesky scratch afterwards)
    int lds_doubles;                  // doubles available for Hc + [S; v'] together (BIG build: decides between the padded and the tight layout)
    int batch;
    const int *rec_count, *rec_list;  // this launch's records (compaction list of the long class), or null
    int *inl_count, *inl_list;        // appended: records whose gate said INLIER
};

constexpr int SGATE_THREADS = 256;         // small build: four waves, one per SIMD, three workgroups per CU
constexpr int SGATE_BIG_THREADS = 768;     // big build (one workgroup per CU by its LDS): three waves per SIMD for the 20 (J, group) items of an 84-row track (512 threads: 70.9 us per 256 records of 21 poses, 768: 66.7, 1024: 87 spilled VGPRs)

// BIG = false: up to 48 rows (three 43 KB workgroups per CU at 10 stereo poses); BIG = true: 49 .. 96 rows (tracks of 13 .. 21 stereo
// poses: 13 .. 21 poses x 4 rows), one workgroup per CU with the whole register file, Hc staged with nrp = 16 TI rows per column where that
// fits 160 KB together with [S; v'] and with 84 rows (TIGHT) where it does not (84 rows: 99.5 + 58.5 KB).
template <bool BIG>
__device__ __forceinline__ void sparse_gate_kernel_body(const SparseGateArgs &a, const int b_in)
{
    // (the record index may come out of a compaction list, i.e. a per-lane load: on the scalar unit every base address below is
    //  uniform and the P gather becomes scalar base + 32-bit lane offset)
    const int b = __builtin_amdgcn_readfirstlane(b_in);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x;
    constexpr int NTH = BIG ? SGATE_BIG_THREADS : SGATE_THREADS;
    PHASE_STAMP(12);
    // (flag, shape, column list and residual requested together: one round trip instead of three dependent ones)
    const unsigned char act_v = a.active[b];
    const int nr_v = a.nr_rec ? a.nr_rec[b] : a.nr;
    int my_acol_raw = a.acol[(size_t)b * a.na_max + min(t, a.na_max - 1)];
    double my_v_raw = a.v[(size_t)b * a.nr + min(t, a.nr - 1)];
    asm volatile("" : "+v"(my_acol_raw), "+v"(my_v_raw));
    if (!act_v) return;                                        // status stays NOT_COMPUTED (preset by the prepare launch)
    const int nr = __builtin_amdgcn_readfirstlane(nr_v);
    if (nr < 2 || nr > a.nr) return;
    const int n = a.n, npose = nr / (2 * a.ncam), na = 7 * npose + 1, na4 = (na + 3) & ~3;
    const int ti = BIG ? max((nr + 15) >> 4, 4) : (nr + 15) >> 4;       // (the BIG build starts at the 4-tile instantiation)
    int Rs = nr + 1;
    while ((Rs & 31) != 15 && (Rs & 31) != 17) Rs++;
    const bool tight = BIG && (size_t)na4 * 16 * ti + (size_t)Rs * nr > (size_t)a.lds_doubles;    // (uniform) does the padded layout fit?
    if (tight) Rs = nr + 1 + ((nr + 1) & 1 ? 0 : 1);            // an odd stride is all the LDS budget allows
    if (tight && nr > HV_GATE_TIGHT_ROWS) return;              // (the launcher sizes LDS for <= 84 rows: never taken)
    const int nrp = tight ? HV_GATE_TIGHT_ROWS : 16 * ti;
    double *Hs = smem, *T = smem + a.hs_doubles;
    int *s_acol = reinterpret_cast<int *>(T + (size_t)Rs * nr);
    const double *Hc = a.Hc + (size_t)b * a.nr * a.na_max;
    // stage Hc k-major with zero padding (rows >= nr, columns >= na), [S; v'] zeroed with the residual row in place
    // (8 elements per thread in flight: a load -> LDS store loop without unrolling waits one HBM / L2 round trip per iteration -- 14 of
    // them at 10 stereo poses, which made this staging half of the kernel)
    const int my_acol = t < na ? my_acol_raw : 0;
    const double my_v = t < nr ? my_v_raw : 0.0;
    const int total = na4 * nrp;
    const unsigned inv_nrp = (unsigned)((0x100000000ull + (unsigned)nrp - 1) / (unsigned)nrp);       // i / nrp = umulhi(i, ceil(2^32 / nrp))
    for (int base = 0; base < total; base += 8 * NTH) {
        double hv_[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + q * NTH + t, u = (int)__umulhi((unsigned)i, inv_nrp), r = i - u * nrp;
            const bool live = i < total && u < na && r < nr;
            hv_[q] = live ? Hc[(size_t)(live ? u : 0) * nr + (live ? r : 0)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = base + q * NTH + t;
            if (i < total) Hs[i] = hv_[q];
        }
    }
    for (int i = t; i < Rs * nr; i += NTH) T[i] = 0.0;             // (row nr of column c is rewritten below by thread c: same thread order
    if (t < na) s_acol[t] = my_acol;                                           //  is not guaranteed, so the barrier comes first)
    __syncthreads();
    if (t < nr) T[(size_t)t * Rs + nr] = my_v;
    __syncthreads();
    PHASE_STAMP(13);
    const double *Pb = a.P + (size_t)b * n * n;
    double chi;
    if constexpr (!BIG) {
        if (ti == 1)      chi = sparse_gate<1, NTH, true>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
        else if (ti == 2) chi = sparse_gate<2, NTH, true>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
        else              chi = sparse_gate<3, NTH, true>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
    } else {
        if (ti <= 4)      chi = sparse_gate<4, NTH, false>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
        else if (ti == 5) chi = sparse_gate<5, NTH, false>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
        else if (!tight)  chi = sparse_gate<6, NTH, false>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
        else              chi = sparse_gate<6, NTH, false, true>(Pb, n, s_acol, na, Hs, T, Rs, nr, a.rd, a.noise_scale, Hs, BIG ? g_phase_stamp : nullptr);
    }
    if (t == 0) {
        const bool broken = !(chi < 1e300);                    // non-positive pivot: reported as CHI2 (ekf_update_kernel phase D)
        const int outlier = broken || ((nr < HV_CHI2INV95_N) ? (chi > d_chi2inv95[nr]) : 0);
        a.status[b] = outlier ? 3 /*CHI2*/ : 0 /*INLIER*/;
        if (!outlier && a.inl_list) a.inl_list[atomicAdd(a.inl_count, 1)] = b;
        if (a.chi2) a.chi2[b] = chi;
    }
}

__global__ __launch_bounds__(SGATE_THREADS, 3) void ekf_sparse_gate_kernel(SparseGateArgs a) { sparse_gate_kernel_body<false>(a, blockIdx.x); }
__global__ __launch_bounds__(SGATE_BIG_THREADS, 1) void ekf_sparse_gate_big_kernel(SparseGateArgs a)
{
    int b = blockIdx.x;
    if (a.rec_list) { if (b >= *a.rec_count) return; b = a.rec_list[b]; }
    sparse_gate_kernel_body<true>(a, b);
}

// ---------------------------------------------------------------------------------------------
// pose augmentation / undo (ekf.cpp:848-903) and housekeeping
// ---------------------------------------------------------------------------------------------
struct AugmentArgs {
    int n, cam_poses, map_dim;
    double *m, *P, *P1, *m1;          // P1/m1: per-filter scratch (n*n, n)
    const int *dropped;               // per filter (or null -> dropped0), -1 = last
    int dropped0;
    double q_pos, q_ori, rd;          // visAugQ diagonal (scaled), augmentR * noiseScale
    const unsigned char *active;
};

__device__ __forceinline__ int aug_src(int i, int dropped, int n)   // visAugA[dropped] as a gather (ekf.cpp:230-248)
{
    if (i < CAM) return i;
    if (i < CAM + POSE) return -1;                                    // slot 0 is re-created by the update
    if (i < CAM + (dropped + 1) * POSE) return i - POSE;
    return i;
}
__device__ __forceinline__ int augh_plus(int i) { return i < 3 ? POS + i : ORI + (i - 3); }   // visAugH (ekf.cpp:267-278)
__device__ __forceinline__ int augh_minus(int i) { return CAM + i; }

constexpr int AUG_THREADS = 1024;

// In-wave transpose of a 16 x 16 f64 tile held in the MFMA C layout (lane (rq, c) holds rows rq + 4 q of
// column c) through a 16 x 17 LDS scratch private to the wavefront: DS operations of one wave execute
// in order, so no barrier is involved.
__device__ __forceinline__ void tile_transpose(double (&v)[4], double *scr, int rq, int c)
{
#pragma unroll
    for (int q = 0; q < 4; q++) scr[(rq + 4 * q) * 17 + c] = v[q];
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = scr[c * 17 + rq + 4 * q];
}

// triangular pair index p -> (I <= J)
__device__ __forceinline__ void tri_decode(int p, int &I, int &J)
{
    J = 0;
    while (p > J) { p -= J + 1; J++; }
    I = p;
}

// The kernel reads the covariance from P and writes the result to P1 (the host swaps the two
// afterwards): the shifted matrix A P A' + Q is a gather of P and is never materialised, so the
// covariance crosses HBM once in each direction instead of three + one times.
// SYM (hv_ekf_symmetrize_augment_dev, r04): the covariance is read as (P + P') / 2 -- maintainPositiveSemiDefinite (ekf.cpp:1059-1067),
// which the backend calls at the end of the visual updates (backend.cpp:1267), folded into the augmentation that follows it: the
// mirrored element of every gathered value belongs to the tile pair the same wavefront reads anyway (L2 hits), and a separate
// symmetrise launch is one more read and write of every covariance. Same values as the two calls in sequence, bit for bit.
template <bool SYM>
__global__ __launch_bounds__(AUG_THREADS) void ekf_augment_kernel(AugmentArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, t = threadIdx.x, n = a.n;
    double *m = a.m + (size_t)b * n;
    const double *P = a.P + (size_t)b * n * n;
    double *Pout = a.P1 + (size_t)b * n * n, *m1 = a.m1 + (size_t)b * n;
    if (a.active && !a.active[b]) {                                  // untouched filter: carry P over to the new buffer
        for (int e = t; e < n * n; e += AUG_THREADS) {
            if (SYM) { const int i = e % n, j = e / n; Pout[e] = 0.5 * (P[e] + P[(size_t)i * n + j]); }
            else Pout[e] = P[e];
        }
        return;
    }
    int dropped = a.dropped ? a.dropped[b] : a.dropped0;
    if (dropped < 0) dropped = a.cam_poses - 1;
    // [HP | K | G]: 7 x n each, row-major [k * n + j] and contiguous: rows 0..13 are the MFMA A operand
    // (HP; K), rows 7..20 the B operand (K; G) of step 4
    double *HP = smem, *K = HP + POSE * n, *G = K + POSE * n;
    double *S0 = G + POSE * n, *Lc = S0 + POSE * POSE, *vres = Lc + POSE * POSE;
    double *scr_all = vres + POSE + 1;                                // 16 waves x 16 x 17 transpose scratch

    // P1 = A P A' + Q as a function (ekf.cpp:230-248, 848-871)
    auto p1 = [&](int i, int j) -> double {
        const int si = aug_src(i, dropped, n), sj = aug_src(j, dropped, n);
        double v = 0.0;
        if (si >= 0 && sj >= 0) {
            v = P[(size_t)sj * n + si];
            if (SYM) v = 0.5 * (v + P[(size_t)si * n + sj]);
        }
        if (i == j && i >= CAM && i < CAM + POSE) v += (i < CAM + 3) ? a.q_pos : a.q_ori;
        return v;
    };

    // 1. m1 = A m
    for (int i = t; i < n; i += AUG_THREADS) { const int s = aug_src(i, dropped, n); m1[i] = s >= 0 ? m[s] : 0.0; }
    // 2. HP = H P1 (7 x n), S0 = HP H'
    for (int e = t; e < POSE * n; e += AUG_THREADS) {
        const int k = e / n, j = e % n;
        HP[e] = p1(augh_plus(k), j) - p1(augh_minus(k), j);
    }
    __syncthreads();
    if (t < POSE * POSE) { const int i = t / POSE, c = t % POSE; S0[t] = HP[i * n + augh_plus(c)] - HP[i * n + augh_minus(c)]; }
    if (t >= 64 && t < 64 + POSE) { const int i = t - 64; vres[i] = -(m1[augh_plus(i)] - m1[augh_minus(i)]); }
    __syncthreads();
    if (t == 0) {                                                   // Cholesky of S = S0 + rd I (7 x 7)
        for (int i = 0; i < POSE; i++) for (int c = 0; c < POSE; c++) Lc[i * POSE + c] = 0.0;
        for (int c = 0; c < POSE; c++) {
            double d = S0[c * POSE + c] + a.rd;
            for (int p = 0; p < c; p++) d -= Lc[c * POSE + p] * Lc[c * POSE + p];
            d = sqrt(d);
            Lc[c * POSE + c] = d;
            for (int i = c + 1; i < POSE; i++) {
                double s = 0.5 * (S0[i * POSE + c] + S0[c * POSE + i]);
                for (int p = 0; p < c; p++) s -= Lc[i * POSE + p] * Lc[c * POSE + p];
                Lc[i * POSE + c] = s / d;
            }
        }
    }
    __syncthreads();
    // K(j, :) = S^-1 HP(:, j)   (K = (S^-1 HP)')
    for (int j = t; j < n; j += AUG_THREADS) {
        double x[POSE];
        for (int i = 0; i < POSE; i++) { double s = HP[i * n + j]; for (int p = 0; p < i; p++) s -= Lc[i * POSE + p] * x[p]; x[i] = s / Lc[i * POSE + i]; }
        for (int i = POSE - 1; i >= 0; i--) { double s = x[i]; for (int p = i + 1; p < POSE; p++) s -= Lc[p * POSE + i] * x[p]; x[i] = s / Lc[i * POSE + i]; }
        double dm = 0;
        for (int i = 0; i < POSE; i++) { K[i * n + j] = x[i]; dm += x[i] * vres[i]; }
        m[j] = m1[j] + dm;                                          // m = A m + K (-H A m)
    }
    __syncthreads();
    // 3. G = P1 H' - K S0 - rd K   (n x 7): then P = P1 - K HP - G K' reproduces the Joseph form
    for (int e = t; e < POSE * n; e += AUG_THREADS) {
        const int c = e / n, i = e % n;
        double g = p1(i, augh_plus(c)) - p1(i, augh_minus(c));
        for (int k = 0; k < POSE; k++) g -= K[k * n + i] * S0[k * POSE + c];
        G[e] = g - a.rd * K[c * n + i];
    }
    __syncthreads();
    // 4. Pout = sym(P1 - K HP - G K')   (maintainPositiveSemiDefinite fused, ekf.cpp:872). A wavefront
    // owns a pair of mirrored 16 x 16 tiles; the rank-14 correction of a tile is 4 f64 MFMA steps,
    //   X(j, i) = P1(i, j) - sum_kk Aop(kk, j) Bop(kk, i),   Aop = [HP; K],  Bop = [K; G],
    // accumulated onto the gathered tile, and the two tiles meet through an in-wave LDS transpose, so
    // every global access is a 128-byte row segment.
    {
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, rq = lane >> 4, c = lane & 15;
        double *scr = scr_all + wave * (16 * 17);
        const int tb = (n + 15) >> 4, npairs = tb * (tb + 1) / 2;
        auto corrected_tile = [&](int i0, int j0, double (&x)[4]) {          // x[q] = X(j0 + rq + 4 q, i0 + c)
            double4v acc;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = i0 + c, j = j0 + rq + 4 * q;
                acc[q] = (i < n && j < n) ? p1(i, j) : 0.0;
            }
#pragma unroll
            for (int sx = 0; sx < 4; sx++) {
                const int kk = 4 * sx + rq;
                const double av = HP[(size_t)min(kk, 13) * n + min(j0 + c, n - 1)];          // Aop(kk, j0 + c)
                const double bv = K[(size_t)min(kk, 13) * n + min(i0 + c, n - 1)];           // Bop(kk, i0 + c)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kk < 14 ? -av : 0.0, kk < 14 ? bv : 0.0, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = acc[q];
        };
        for (int p = wave; p < npairs; p += AUG_THREADS / 64) {
            int I, J;
            tri_decode(p, I, J);
            const int i0 = 16 * I, j0 = 16 * J;
            double x[4], y[4];
            corrected_tile(i0, j0, x);                                        // element (i0 + c, j0 + r)
            if (I != J) corrected_tile(j0, i0, y);                            // element (j0 + c, i0 + r)
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = x[q];
            }
            tile_transpose(y, scr, rq, c);                                    // now the mirror of x[q]
#pragma unroll
            for (int q = 0; q < 4; q++) {
                x[q] = 0.5 * (x[q] + y[q]);
                const int i = i0 + c, j = j0 + rq + 4 * q;
                if (i < n && j < n) Pout[(size_t)j * n + i] = x[q];
            }
            if (I != J) {
                tile_transpose(x, scr, rq, c);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = j0 + c, j = i0 + rq + 4 * q;
                    if (i < n && j < n) Pout[(size_t)j * n + i] = x[q];
                }
            }
        }
    }
    __syncthreads();
    const int nq = 1 + (n - a.map_dim - CAM) / POSE;
    if (t < nq) normalize4(m + (t == 0 ? ORI : CAM + POSE * (t - 1) + 3));
}

struct ShiftArgs { int n, map_dim; double *m, *P, *P1, *m1; const unsigned char *active; };

// updateUndoAugmentation (ekf.cpp:888-903): m = U m, P = U P U' with the shift visUnaugmentA (251-265)
__device__ __forceinline__ int unaug_src(int i, int n, int map_dim)
{
    const int trail = n - map_dim;
    if (i < CAM || i >= trail) return i;
    return (i + POSE < trail) ? i + POSE : -1;
}

// writes the shifted covariance to P1 (the host swaps P and P1 afterwards) and the shifted mean in place
__global__ __launch_bounds__(1024) void ekf_unaugment_kernel(ShiftArgs a)
{
    const int b = blockIdx.x, t = threadIdx.x, n = a.n;
    double *m = a.m + (size_t)b * n;
    const double *P = a.P + (size_t)b * n * n;
    double *P1 = a.P1 + (size_t)b * n * n, *m1 = a.m1 + (size_t)b * n;
    if (a.active && !a.active[b]) {
        for (int e = t; e < n * n; e += 1024) P1[e] = P[e];
        return;
    }
    for (int i = t; i < n; i += 1024) { const int s = unaug_src(i, n, a.map_dim); m1[i] = s >= 0 ? m[s] : 0.0; }
    for (int e = t; e < n * n; e += 1024) {
        const int si = unaug_src(e % n, n, a.map_dim), sj = unaug_src(e / n, n, a.map_dim);
        P1[e] = (si >= 0 && sj >= 0) ? P[(size_t)sj * n + si] : 0.0;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) m[i] = m1[i];
}

// maintainPositiveSemiDefinite (ekf.cpp:1059-1067): P = (P + P') / 2, one wavefront per pair of mirrored
// 16 x 16 tiles (in-wave LDS transpose: both tiles are read and written as 128-byte row segments)
__global__ __launch_bounds__(1024) void ekf_symmetrize_kernel(int n, double *Pall)
{
    __shared__ double scr_all[16 * 16 * 17];
    double *P = Pall + (size_t)blockIdx.x * n * n;
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, rq = lane >> 4, c = lane & 15;
    double *scr = scr_all + wave * (16 * 17);
    const int tb = (n + 15) >> 4, npairs = tb * (tb + 1) / 2;
    for (int p = wave; p < npairs; p += 16) {
        int I, J;
        tri_decode(p, I, J);
        const int i0 = 16 * I, j0 = 16 * J;
        double x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = rq + 4 * q;
            x[q] = (i0 + c < n && j0 + r < n) ? P[(size_t)(j0 + r) * n + i0 + c] : 0.0;
            y[q] = (j0 + c < n && i0 + r < n) ? P[(size_t)(i0 + r) * n + j0 + c] : 0.0;
        }
        tile_transpose(y, scr, rq, c);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            x[q] = 0.5 * (x[q] + y[q]);
            const int r = rq + 4 * q;
            if (i0 + c < n && j0 + r < n) P[(size_t)(j0 + r) * n + i0 + c] = x[q];
        }
        if (I != J) {
            tile_transpose(x, scr, rq, c);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = rq + 4 * q;
                if (j0 + c < n && i0 + r < n) P[(size_t)(i0 + r) * n + j0 + c] = x[q];
            }
        }
    }
}

__global__ void fill_doubles_kernel(double *dst, int n, double value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

__global__ void ekf_normalize_kernel(int n, int map_dim, double *mall, int only_current)
{
    double *m = mall + (size_t)blockIdx.x * n;
    const int nq = only_current ? 1 : 1 + (n - map_dim - CAM) / POSE;
    const int t = threadIdx.x;
    if (t < nq) normalize4(m + (t == 0 ? ORI : CAM + POSE * (t - 1) + 3));
}

// transformTo (ekf.cpp:704-758): m = A m (+ translation of every position), P = A P A' with the
// block-diagonal A = diag(pC, pC, qC, I_10, {pC, qC} x poses [, I]); one thread per block pair.
struct TransformArgs { int n, cam_poses; double *m, *P; double pC[9], qC[16], tr[3]; };   // row-major blocks

__device__ __forceinline__ void blk_of(int idx, int cam_poses, int &start, int &size, int &kind)
{
    // kind 0: identity 1x1, 1: pC (3x3), 2: qC (4x4)
    if (idx == 0) { start = POS; size = 3; kind = 1; return; }
    if (idx == 1) { start = VEL; size = 3; kind = 1; return; }
    if (idx == 2) { start = ORI; size = 4; kind = 2; return; }
    if (idx < 13) { start = BGA + (idx - 3); size = 1; kind = 0; return; }
    const int p = idx - 13;
    if (p < 2 * cam_poses) { start = CAM + POSE * (p >> 1) + ((p & 1) ? 3 : 0); size = (p & 1) ? 4 : 3; kind = (p & 1) ? 2 : 1; return; }
    start = CAM + POSE * cam_poses + (p - 2 * cam_poses); size = 1; kind = 0;
}

__global__ __launch_bounds__(256) void ekf_transform_kernel(TransformArgs a)
{
    const int n = a.n, nblk = 13 + 2 * a.cam_poses + (n - CAM - POSE * a.cam_poses);
    double *P = a.P, *m = a.m;
    for (int e = threadIdx.x + blockIdx.x * 256; e < nblk * nblk; e += 256 * gridDim.x) {
        int si, ni, ki, sj, nj, kj;
        blk_of(e % nblk, a.cam_poses, si, ni, ki);
        blk_of(e / nblk, a.cam_poses, sj, nj, kj);
        if (ki == 0 && kj == 0) continue;
        double X[16], Y[16];
        for (int i = 0; i < ni; i++) for (int j = 0; j < nj; j++) X[i * 4 + j] = P[(size_t)(sj + j) * n + si + i];
        const double *Ai = ki == 1 ? a.pC : a.qC, *Aj = kj == 1 ? a.pC : a.qC;
        for (int i = 0; i < ni; i++) for (int j = 0; j < nj; j++) {      // Y = Ai X
            double s = 0;
            if (ki == 0) s = X[i * 4 + j]; else for (int k = 0; k < ni; k++) s += Ai[i * ni + k] * X[k * 4 + j];
            Y[i * 4 + j] = s;
        }
        for (int i = 0; i < ni; i++) for (int j = 0; j < nj; j++) {      // P = Y Aj'
            double s = 0;
            if (kj == 0) s = Y[i * 4 + j]; else for (int k = 0; k < nj; k++) s += Y[i * 4 + k] * Aj[j * nj + k];
            P[(size_t)(sj + j) * n + si + i] = s;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < nblk) {
        int s, sz, kind;
        blk_of(threadIdx.x, a.cam_poses, s, sz, kind);
        if (kind) {
            double x[4], y[4];
            for (int i = 0; i < sz; i++) x[i] = m[s + i];
            const double *A = kind == 1 ? a.pC : a.qC;
            for (int i = 0; i < sz; i++) { double t2 = 0; for (int k = 0; k < sz; k++) t2 += A[i * sz + k] * x[k]; y[i] = t2; }
            const bool is_pos = kind == 1 && s != VEL;
            for (int i = 0; i < sz; i++) m[s + i] = y[i] + (is_pos ? a.tr[i] : 0.0);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Ekf {
    Ctx *c = nullptr;
    hv_ekf_params par{};
    int batch = 0, n = 0, cam = 0, map_dim = 0;
    double noise_scale = 0;
    double *m = nullptr, *P = nullptr, *P1 = nullptr, *m1 = nullptr, *Q = nullptr, *dydx = nullptr, *ws = nullptr;
    double *sH = nullptr, *sv = nullptr, *sr = nullptr, *schi2 = nullptr, *simu = nullptr;   // staging for host-pointer calls
    int *sstatus = nullptr, *sdrop = nullptr;
    unsigned char *sactive = nullptr;
    size_t sH_cap = 0;
    int max_rows = 0;
    // buffers of hv_ekf_visual_track_dev (row f3), sized on first use
    double *vuH = nullptr, *vuv = nullptr, *vupf = nullptr;
    unsigned char *vuactive = nullptr;
    int *vurows = nullptr;                                // ragged batches: per-filter rows of the current visit (written by vu_prepare)
    int *sprows = nullptr;                                // ... and per (track, filter) record of the speculative loop
    int vu_rows = 0;
    // speculative frame loop: per (track, filter) records + per-filter cursor and the update count each record was prepared at
    double *spH = nullptr, *spv = nullptr, *sppf = nullptr;
    unsigned char *spactive = nullptr;
    int *spcursor = nullptr, *spepoch = nullptr;
    int *spcursor2 = nullptr, *sppub = nullptr;           // fused gate + apply passes: second cursor (ping-pong), published decisions
    size_t sp_records = 0; int sp_rows = 0;
    // device staging of the host-pointer entry hv_ekf_visual_track: idx | features | velocities | y | status | gate | chi2 | pf
    unsigned char *vustage = nullptr;
    size_t vustage_bytes = 0;
    // fused prepare + gate (compact Jacobians live in vuH / spH): the active-column lists of the records
    int *vuacol = nullptr, *spacol = nullptr;
    // long-track classes of a ragged visit (visual_track_dev_impl): own stream, events, Jacobian / residual / active buffers
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;    // fork onto / join of the context's second stream (Ctx::aux_stream) inside a visit
    double *sideH = nullptr, *sidev = nullptr;
    unsigned char *side_active = nullptr;
    int side_rows = 0;
    int *side_acol = nullptr; double *side_dm = nullptr;
    double *tri_rec = nullptr; int tri_stride = 0;        // factor records of vu_tri_kernel (split form of a visit, r06): [batch][tri_stride]
    int *err_dev = nullptr;                               // device error word (UpdateArgs::err)
    double *bH = nullptr, *bv = nullptr; int *brows = nullptr; unsigned char *bany = nullptr; int b_rows = 0;   // batchVisualUpdate: stacked [H; v], rows, flags
    double *gate_scale = nullptr;                         // [batch] per-filter multiplier of the outlier thresholds inside a frame loop (backend.cpp:1192-1193)
    bool gate_scale_on = false;                           // set by the frame loop while its visits run with a growth factor != 1
    int *visit_counts = nullptr, *visit_lists = nullptr;  // compaction lists of a visit: counts {inliers short, long records, inliers long}, lists 3 x [batch]
    // the counts exist once per visit of a frame loop (VISIT_SLOTS x 4 ints, zeroed by ONE memset per frame; visit_slot = the running
    // visit, set by the loop) plus one set for stand-alone visits (zeroed per call): a memset node per visit was 20 more graph nodes
    static constexpr int VISIT_SLOTS = 64;
    int visit_slot = -1;
    int *visit_order = nullptr;                           // [VISIT_SLOTS][batch] launch_visit_order of the running frame loop, valid while visit_order_ok
    int *visit_long = nullptr, *visit_long_count = nullptr;   // ... its long-class lists [VISIT_SLOTS][batch] and their lengths [VISIT_SLOTS]
    bool visit_order_ok = false;
};

// compact-H description handed to ekf_launch_update (null acol: dense H of l columns); half / nr_full / dm: block update of a long
// track (UpdateArgs::half)
struct CompactH { const int *acol = nullptr; int na_max = 0, ncam = 1; int half = 0, nr_full = 0; double *dm = nullptr; const int *rec_count = nullptr, *rec_list = nullptr; int *gate_rw = nullptr;
                  int half_auto = 0; int *sel_io = nullptr; int *epoch = nullptr; };

// an update launch prepared but not issued (ekf_launch_update's `defer`): two of them can share one grid (ekf_launch_update_dual)
struct UpdateLaunch { UpdateArgs a; size_t base_bytes = 0; int kmode = -1, ti = 0, lbk = 0; };

static int ekf_launch_update(Ekf *e, int nr, int l, const double *H_dev, const double *v_dev, const double *rdiag_dev,
                             double rd0, int mode, int generic, int normalize_all, double *chi2_dev, int *status_dev,
                             const unsigned char *active_dev, const int *require_inlier_dev = nullptr,
                             int *success_counter_dev = nullptr, double rd1 = 0.0, bool *two_r_done = nullptr,
                             int spec = 0, int n_tracks = 0, int *cursor_dev = nullptr, int max_successful = 0,
                             const int *gate_in_dev = nullptr, int *cursor_out_dev = nullptr, int *pub_dev = nullptr, int pass_id = 0,
                             const int *nr_rec_dev = nullptr, const CompactH *compact = nullptr, int nr_stride = 0, UpdateLaunch *defer = nullptr)
{
    // nr_stride (ragged launches that serve one length class): rows of the LONGEST record of the batch = the record stride of H and v;
    // nr is then the most rows this launch processes (kernel variant, LDS carve), longer records are skipped by their `active` flag
    Ctx *c = e->c;
    if (nr_stride <= 0) nr_stride = nr;
    if (nr < 1 || nr > e->max_rows || l < 1 || l > e->n) return HV_ERR_INVALID;
    // the chi2 gate needs chi2inv95[nr] (the reference asserts n < chi2inv95.size(): ekf.cpp:806); mode 1 with an
    // inlier requirement is the update half of a gate that already ran
    if (!generic && mode != 1 && nr >= HV_CHI2INV95_N) return HV_ERR_INVALID;
    UpdateArgs a{};
    a.n = e->n; a.nr = nr; a.l = l; a.R = nr + e->n + 1;
    // LDS stride of T: 15 or 17 mod 32 doubles. Odd keeps the 16 lanes of a row-strided operand read (S = HP H')
    // in distinct banks; 2 Rs = 30 or 34 mod 64 dwords puts the k-groups of a column-strided read (the
    // Cholesky panels, Y'Y) half a bank array apart.
    int r_pad = a.R;
    while ((r_pad & 31) != 15 && (r_pad & 31) != 17) r_pad++;
    a.Rs = r_pad;
    a.mode = mode; a.generic = generic; a.normalize_all = normalize_all; a.map_dim = e->map_dim;
    a.m = e->m; a.P = e->P; a.H = H_dev; a.v = v_dev; a.rdiag = rdiag_dev; a.rd0 = rd0; a.rd1 = rd1; a.noise_scale = e->noise_scale;
    a.ws = e->ws; a.chi2 = chi2_dev; a.status = status_dev; a.active = active_dev; a.require_inlier = require_inlier_dev; a.success_counter = success_counter_dev;
    a.spec = spec; a.n_tracks = n_tracks; a.cursor = cursor_dev; a.max_successful = max_successful; a.gate_in = gate_in_dev;
    a.cursor_out = cursor_out_dev; a.pub = pub_dev; a.pass_id = pass_id; a.nr_rec = nr_rec_dev; a.err = e->err_dev;
    a.h_stride = (size_t)nr_stride * l; a.v_stride = nr_stride;
    if (compact && compact->acol) {
        a.acol = compact->acol; a.na_max = compact->na_max; a.ncam = compact->ncam; a.h_stride = (size_t)nr_stride * compact->na_max;
        a.half = compact->half; a.nr_full = compact->nr_full;
        if (a.half == 1) { a.dm_out = compact->dm; a.gate_rw = compact->gate_rw; }
        if (a.half == 2) a.dm_in = compact->dm;
        a.rec_count = compact->rec_count; a.rec_list = compact->rec_list;
        a.half_auto = compact->half_auto; a.sel_io = compact->sel_io; a.epoch = compact->epoch;
        if (a.half_auto && (spec != 2 || !a.half || !a.sel_io || !a.dm_out == !a.dm_in)) return HV_ERR_INVALID;
    }
    size_t tall = (((size_t)a.Rs * nr + 1) & ~(size_t)1) * sizeof(double);
    const size_t small = (size_t)(256 + 544 + UPD_THREADS / 64 + 2) * sizeof(double);           // W + col (incl. dump area) + red + flag
    const int ti = (nr + 15) / 16, lbk = (l + 15) / 16;
    const size_t hbytes = (size_t)(16 * ti) * (16 * lbk) * sizeof(double);                      // zero-padded H
    const size_t lds_cap = 160 * 1024;                       // the whole LDS of a CU: T alone reaches 154 KB at 80 rows (20 stereo poses)
    a.use_lds = tall + small <= lds_cap;
    if (!a.use_lds) { a.Rs = a.R; tall = (((size_t)a.R * nr + 1) & ~(size_t)1) * sizeof(double); }   // global workspace: no padding
    int kmode = !a.use_lds ? 0 : (e->n <= 160 && nr <= 48 && tall + small + hbytes <= lds_cap) ? 2 : 1;
    // HV_EKF_GATE_KMODE (environment, experiments only): kernel variant for gate-only launches (1 = H streamed from L2, 2 WGs / CU)
    if (c->knob.ekf_gate_kmode == 1 && mode == 0 && !spec && kmode == 2) kmode = 1;
    if (a.acol && kmode != 2) return HV_ERR_UNSUPPORTED; // compact H is staged by the LDS-resident kernel only (vu_fused_supported)
    if (mode == 3) {                                     // gate (rd0) + update (rd1) in one launch: MODE 2 kernels only
        const bool can = kmode == 2 && ((size_t)(nr + 1) * nr + 256) * sizeof(double) <= hbytes;
        if (two_r_done) *two_r_done = can;
        if (!can) return HV_OK;                          // the caller falls back to two launches
    }
    if (spec && kmode != 2) return HV_ERR_UNSUPPORTED;   // the speculative loop keeps every record in the LDS-resident kernel
    const size_t shmem = kmode == 2 ? tall + small + hbytes : kmode == 1 ? tall + small : small;
    a.batch = e->batch;
    if (defer) {
        if (mode == 3 || spec) return HV_ERR_INVALID;
        defer->a = a; defer->base_bytes = tall + small; defer->kmode = kmode; defer->ti = ti; defer->lbk = lbk;
        return HV_OK;
    }
    using Kern = void (*)(UpdateArgs);
    if (a.half_auto && (kmode != 2 || ti != 3)) return HV_ERR_INVALID;
    const Kern kern = a.half_auto ? (Kern)ekf_update_spec_long_kernel : kmode == 0 ? (Kern)ekf_update_kernel<0, 0> : kmode == 1 ? (Kern)ekf_update_kernel<1, 0>
                    : ti == 1 ? (Kern)ekf_update_kernel<2, 1> : ti == 2 ? (Kern)ekf_update_kernel<2, 2> : (Kern)ekf_update_kernel<2, 3>;
    // the attribute is per device: one flag per device ordinal (several contexts / devices may live in one process)
    static bool attr_set_dev[64] = {};
    bool &attr_set = attr_set_dev[c->p.device & 63];
    if (!attr_set) {
        for (Kern k : { (Kern)ekf_update_kernel<1, 0>, (Kern)ekf_update_kernel<2, 1>, (Kern)ekf_update_kernel<2, 2>, (Kern)ekf_update_kernel<2, 3>, (Kern)ekf_update_spec_long_kernel })
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    ScopedKernelTime tm(c, HV_K_EKF_UPDATE);
    hipLaunchKernelGGL(kern, dim3(e->batch, (spec == 1 || spec == 3) ? n_tracks : 1), dim3(UPD_THREADS), shmem, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

// two deferred masked MODE 2 launches over disjoint filters as one grid; *done = false when their shapes do not allow it (the caller
// then issues them one after the other)
static int ekf_launch_update_dual(Ekf *e, const UpdateLaunch &A, const UpdateLaunch &B, int part, bool *done)
{
    Ctx *c = e->c;
    *done = false;
    if (A.kmode != 2 || B.kmode != 2 || !A.a.rec_list || !B.a.rec_list || !A.a.rec_count || !B.a.rec_count) return HV_OK;
    const int ti = A.ti > B.ti ? A.ti : B.ti;
    if (ti != 3) return HV_OK;                                     // (the only pairing the visit loop produces: 44 + 42 rows)
    const size_t sa = A.base_bytes + (size_t)(16 * ti) * (16 * A.lbk) * sizeof(double), sb = B.base_bytes + (size_t)(16 * ti) * (16 * B.lbk) * sizeof(double);
    const size_t shmem = sa > sb ? sa : sb;
    if (shmem > 160 * 1024) return HV_OK;
    static bool attr_set_dev[64] = {};
    bool &attr_set = attr_set_dev[c->p.device & 63];
    if (!attr_set) {
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(ekf_update_dual_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    ScopedKernelTime tm(c, HV_K_EKF_UPDATE);
    UpdatePair pair;
    pair.a[0] = A.a; pair.a[1] = B.a;
    hipLaunchKernelGGL((ekf_update_dual_kernel<2, 3>), dim3(e->batch), dim3(UPD_THREADS), shmem, c->stream, pair, part, c->num_cus);
    HV_HIP(c, hipGetLastError());
    *done = true;
    return HV_OK;
}

static int ekf_launch_gate_stream(Ekf *e, int nr, int l, const double *H_dev, const double *v_dev, double rd, double *chi2_dev,
                                  int *status_dev, const unsigned char *active_dev, const int *success_counter_dev, int max_successful,
                                  bool *done)
{
    Ctx *c = e->c;
    *done = false;
    if (nr < 1 || nr > 48 || nr >= HV_CHI2INV95_N || l < 1 || l > e->n) return HV_OK;        // the caller keeps its other route
    GateArgs a{};
    a.n = e->n; a.nr = nr; a.l = l;
    int r_pad = nr + 1;
    while ((r_pad & 31) != 15 && (r_pad & 31) != 17) r_pad++;
    a.Rs = r_pad;
    a.P = e->P; a.H = H_dev; a.v = v_dev; a.rd = rd; a.noise_scale = e->noise_scale; a.chi2 = chi2_dev; a.status = status_dev;
    a.active = active_dev; a.success_counter = success_counter_dev; a.max_successful = max_successful;
    const int ti = (nr + 15) / 16, lbk = (l + 15) / 16;
    const size_t hs = (size_t)(16 * ti) * (16 * lbk), shmem = sizeof(double) * (hs + (size_t)a.Rs * nr + 2);
    if (hs < 256 + 544 + 8) return HV_OK;                          // W / col / red borrow the H area
    using Kern = void (*)(GateArgs);
    const Kern kern = ti == 1 ? (Kern)ekf_gate_stream_kernel<1> : ti == 2 ? (Kern)ekf_gate_stream_kernel<2> : (Kern)ekf_gate_stream_kernel<3>;
    static bool attr_set_dev[64] = {};
    bool &attr_set = attr_set_dev[c->p.device & 63];
    if (!attr_set) {
        for (Kern k : { (Kern)ekf_gate_stream_kernel<1>, (Kern)ekf_gate_stream_kernel<2>, (Kern)ekf_gate_stream_kernel<3> })
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    if (shmem > 96 * 1024) return HV_OK;
    ScopedKernelTime tm(c, HV_K_EKF_UPDATE);
    hipLaunchKernelGGL(kern, dim3(e->batch), dim3(GATE_THREADS), shmem, c->stream, a);
    HV_HIP(c, hipGetLastError());
    *done = true;
    return HV_OK;
}

// ---------------------------------------------------------------------------------------------
// Hybrid map (backend.cpp:1160-1168): an inlier pose-trail track that is OFFERED a map slot becomes a map point -- insertMapPoint
// (ekf.cpp:911-921: the slot's rows and columns of P zeroed, 1e6 on its diagonal, the triangulated point in the mean) -- INSTEAD of
// being applied. One workgroup per filter; active_out = the filters whose track still takes updateVisualTrack.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ekf_hybrid_insert_kernel(int n, int map_base, double *m_all, double *P_all, const unsigned char *active,
                                                                const int *gate, const int *map_index, const int *offer, const double *pf,
                                                                unsigned char *active_out)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = active[b] != 0, inlier = act && gate[b] == 0;
    const int slot = (inlier && (!map_index || map_index[b] < 0) && offer) ? offer[b] : -1;
    if (t == 0) active_out[b] = (act && slot < 0) ? 1 : 0;
    if (slot < 0) return;
    const int off = map_base + 3 * slot;
    double *m = m_all + (size_t)b * n, *P = P_all + (size_t)b * n * n;
    for (int i = t; i < 3 * n; i += 256) {
        const int k = i / n, j = i - k * n;
        P[(size_t)(off + k) * n + j] = 0.0;
        P[(size_t)j * n + off + k] = 0.0;
    }
    __syncthreads();
    if (t < 3) { P[(size_t)(off + t) * n + off + t] = 1e6; m[off + t] = pf[3 * b + t]; }
}

// ---------------------------------------------------------------------------------------------
// batchVisualUpdate (backend.cpp:1001-1010, 1169-1183, 1255-1262): inside a batch every track is gated against the SAME (m, P) -- the
// updates are deferred --, so one speculative pass (every pending track prepared and gated in one launch) IS the reference's loop up to
// the next flush. This kernel is the bookkeeping of that loop for one filter: it walks the pending tracks in visit order, appends the
// inliers' blocks [H; v] to the filter's batch while they fit max_rows and the quota lasts, and leaves the stacked dense H (column-major,
// leading dimension = the batch's row count) and v for ONE updateVisualTrack launch (ragged dense update: rows_out). A block that does
// not fit ends the pass: the batch is flushed by that update, the block opens the NEXT batch unchanged (carry), and the next pass gates
// the tracks behind it against the updated state -- the reference's flush-then-append.
// ---------------------------------------------------------------------------------------------
struct BatchAssembleArgs {
    int n, n_tracks, batch, max_rows, max_successful, rows_stride, na_max, ncam;
    const double *Hc, *v;             // compact records of the gate launch: [n_tracks][batch] x (rows_stride x na_max), x rows_stride
    const int *acol, *nr_rec;         // [records][na_max]; rows of every record (null: rows_stride)
    const unsigned char *active; const int *gate;
    int *cursor, *success_counter, *carry;   // carry [batch]: the track whose block opens the next batch (-1: none)
    double *Hd, *vd;                  // [batch][max_rows x n], [batch][max_rows]
    int *rows_out; unsigned char *any_out;
};
constexpr int BATCH_THREADS = 256;
__global__ __launch_bounds__(BATCH_THREADS) void ekf_batch_assemble_kernel(BatchAssembleArgs a)
{
    __shared__ int s_list[64], s_off[64], s_cnt, s_rows;
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) {
        int succ = a.success_counter[b], rows = 0, cnt = 0, j = a.cursor[b];
        // the track whose block did not fit the previous batch opens this one AS IT WAS prepared and gated -- against the state before
        // the flush: the reference has its H, f, y in hand when it flushes and appends them afterwards (backend.cpp:1171-1182)
        const int cj = a.carry[b];
        if (cj >= 0) {
            const int rec = cj * a.batch + b;
            s_list[0] = cj; s_off[0] = 0; rows = a.nr_rec ? a.nr_rec[rec] : a.rows_stride; cnt = 1; ++succ;
            a.carry[b] = -1;
        }
        for (; j < a.n_tracks; ++j) {
            if (succ >= a.max_successful) break;                                   // quota (backend.cpp:1233): the next pass marks the rest NOT_VISITED
            const int rec = j * a.batch + b;
            if (!a.active[rec] || a.gate[rec] != 0) continue;                     // failed triangulation / outlier: final, nothing to apply
            const int nr = a.nr_rec ? a.nr_rec[rec] : a.rows_stride;
            if (rows + nr > a.max_rows || cnt == 64) { a.carry[b] = j; ++j; break; }      // flush first; the block waits for the next batch
            s_list[cnt] = j; s_off[cnt] = rows; rows += nr; ++cnt; ++succ;
        }
        a.cursor[b] = j;                                                           // (n_tracks when every pending track is final)
        a.success_counter[b] = succ;
        a.rows_out[b] = rows; a.any_out[b] = rows > 0 ? 1 : 0;
        s_cnt = cnt; s_rows = rows;
    }
    __syncthreads();
    const int cnt = s_cnt, rows = s_rows;
    if (rows == 0) return;
    double *Hd = a.Hd + (size_t)b * a.max_rows * a.n, *vd = a.vd + (size_t)b * a.max_rows;
    for (int i = t; i < rows * a.n; i += BATCH_THREADS) Hd[i] = 0.0;
    __syncthreads();
    for (int q = 0; q < cnt; ++q) {
        const int rec = s_list[q] * a.batch + b, off = s_off[q];
        const int nr = a.nr_rec ? a.nr_rec[rec] : a.rows_stride, na = 7 * (nr / (2 * a.ncam)) + 1;
        const double *Hc = a.Hc + (size_t)rec * a.rows_stride * a.na_max, *v = a.v + (size_t)rec * a.rows_stride;
        const int *acol = a.acol + (size_t)rec * a.na_max;
        for (int i = t; i < na * nr; i += BATCH_THREADS) {
            const int u = i / nr, r = i - u * nr;
            Hd[(size_t)acol[u] * rows + off + r] = Hc[i];                          // (compact column u of the record, leading dimension nr)
        }
        for (int r = t; r < nr; r += BATCH_THREADS) vd[off + r] = v[r];
    }
}

// ekf_sparse_gate_kernel over the compact records of a prepare launch (np = poses of the longest record)
static int ekf_launch_sparse_gate(Ekf *e, int np, int ncam, const double *Hc_dev, const double *v_dev, const int *acol_dev, const int *nr_rec_dev,
                                  const unsigned char *active_dev, double rd, double *chi2_dev, int *status_dev,
                                  const int *rec_count = nullptr, const int *rec_list = nullptr, int *inl_count = nullptr, int *inl_list = nullptr,
                                  hipStream_t stream = nullptr)
{
    Ctx *c = e->c;
    if (!stream) stream = c->stream;
    const int nr = 2 * np * ncam, na_max = 7 * np + 1, na4 = (na_max + 3) & ~3, nrp = 16 * ((nr + 15) / 16);
    if (nr < 2 || nr > 96 || nr >= HV_CHI2INV95_N || !active_dev || !status_dev) return HV_ERR_INVALID;
    const bool big = nr > 48;
    SparseGateArgs a{};
    a.n = e->n; a.nr = nr; a.ncam = ncam; a.na_max = na_max; a.P = e->P; a.Hc = Hc_dev; a.v = v_dev; a.acol = acol_dev; a.nr_rec = nr_rec_dev;
    a.active = active_dev; a.rd = rd; a.noise_scale = e->noise_scale; a.chi2 = chi2_dev; a.status = status_dev;
    int Rs = nr + 1;
    while ((Rs & 31) != 15 && (Rs & 31) != 17) Rs++;
    // LDS: Hc staged [na4][nrp] + [S; v'] (Rs x nr) + the column list. The launch is sized for its longest record; in the big build a
    // record whose padded layout does not fit (84 rows) uses the tight one (nrp = 84, odd Rs) inside the same carve.
    size_t hs = (size_t)na4 * nrp, tt = (size_t)Rs * nr;
    // (the gate's turn counters are 32 bytes of STATIC LDS: the dynamic part of the big build ends 64 bytes below the CU's 160 KB)
    constexpr size_t BIG_CAP = 160 * 1024 - 64;
    const size_t cap = big ? BIG_CAP : (size_t)96 * 1024, ints = sizeof(int) * (size_t)(na_max + 2);
    if (big && sizeof(double) * (hs + tt) + ints > cap) {
        if (nr > HV_GATE_TIGHT_ROWS) return HV_ERR_UNSUPPORTED;
        hs = (size_t)na4 * HV_GATE_TIGHT_ROWS; tt = (size_t)(nr + 2) * nr;
    }
    if (hs < 824) hs = 824;
    a.hs_doubles = (int)hs; a.lds_doubles = (int)(hs + tt);
    const size_t shmem = sizeof(double) * (hs + tt) + ints;
    if (shmem > cap) return HV_ERR_UNSUPPORTED;
    static bool attr_set_dev[64] = {};
    bool &attr_set = attr_set_dev[c->p.device & 63];
    if (!attr_set) {
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(ekf_sparse_gate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(ekf_sparse_gate_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BIG_CAP));
        attr_set = true;
    }
    ScopedKernelTime tm(c, HV_K_EKF_GATE, stream);
    a.rec_count = rec_count; a.rec_list = rec_list; a.inl_count = inl_count; a.inl_list = inl_list;
    a.batch = e->batch;
    if (big) hipLaunchKernelGGL(ekf_sparse_gate_big_kernel, dim3((unsigned)e->batch), dim3(SGATE_BIG_THREADS), shmem, stream, a);
    else     hipLaunchKernelGGL(ekf_sparse_gate_kernel, dim3(e->batch), dim3(SGATE_THREADS), shmem, stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace hv

using hv::Ctx;
using hv::Ekf;

extern "C" {

struct hv_ekf { Ekf e; };

void hv_ekf_default_params(hv_ekf_params *p)
{
    if (!p) return;
    p->cameraTrailLength = 20; p->hybridMapSize = 0;
    p->noiseScale = 100; p->gravity = 9.819; p->augmentR = 1e-9; p->initZuptR = 1e-4; p->rotationZuptR = 1e-6;
    p->noiseInitialPos = 1e-5; p->noiseInitialOri = 0.0316227766; p->noiseInitialVel = 0.1;
    p->noiseInitialPosTrail = 100; p->noiseInitialOriTrail = 3.16227766;
    p->noiseInitialBGA = 1e-3; p->noiseInitialBAA = 1e-6; p->noiseInitialBAT = 1e-5; p->noiseInitialSFT = 1e-5;
    p->noiseProcessAcc = 0.003; p->noiseProcessGyro = 0.00017; p->noiseProcessBAA = 1e-4; p->noiseProcessBGA = 0;
    p->noiseProcessBAARev = 0.1; p->noiseProcessBGARev = 0.1;
}

void hv_ekf_destroy(hv_ekf *h)
{
    if (!h) return;
    Ekf *e = &h->e;
    if (e->c && e->c->stream) (void)hipStreamSynchronize(e->c->stream);
    void *ptrs[] = { e->m, e->P, e->P1, e->m1, e->Q, e->dydx, e->ws, e->sH, e->sv, e->sr, e->schi2, e->simu,
                     e->sstatus, e->sdrop, e->sactive, e->vuH, e->vuv, e->vupf, e->vuactive, e->vustage,
                     e->spH, e->spv, e->sppf, e->spactive, e->spcursor, e->spepoch, e->spcursor2, e->sppub, e->vurows, e->sprows,
                     e->vuacol, e->spacol, e->err_dev, e->gate_scale, e->bH, e->bv, e->brows, e->bany, e->sideH, e->sidev, e->side_active, e->side_acol, e->side_dm, e->tri_rec, e->visit_counts, e->visit_lists, e->visit_order, e->visit_long, e->visit_long_count };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (e->c && e->c->aux_stream) (void)hipStreamSynchronize(e->c->aux_stream);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    delete h;
}

int hv_ekf_create(hv_ctx *ctx, const hv_ekf_params *par, int batch, hv_ekf **out)
{
    if (!ctx || !par || !out || batch < 1 || par->cameraTrailLength < 1 || par->hybridMapSize < 0) return HV_ERR_INVALID;
    *out = nullptr;
    Ctx *c = hv::ctx_of(ctx);
    hv_ekf *h = new (std::nothrow) hv_ekf();
    if (!h) return HV_ERR_NOMEM;
    Ekf *e = &h->e;
    e->c = c; e->par = *par; e->batch = batch;
    e->cam = par->cameraTrailLength; e->map_dim = 3 * par->hybridMapSize;
    const int n = e->n = hv::INER + hv::POSE * e->cam + e->map_dim;
    e->noise_scale = par->noiseScale * par->noiseScale;
    e->max_rows = n;
    const size_t nn = (size_t)n * n;
    bool ok = true;
    auto alloc = [&](auto &ptr, size_t bytes) { if (ok && hipMalloc(reinterpret_cast<void **>(&ptr), bytes) != hipSuccess) ok = false; };
    alloc(e->m, sizeof(double) * n * batch); alloc(e->P, sizeof(double) * nn * batch);
    alloc(e->P1, sizeof(double) * nn * batch); alloc(e->m1, sizeof(double) * n * batch);
    alloc(e->Q, sizeof(double) * 144 * batch); alloc(e->dydx, sizeof(double) * 400 * batch);
    alloc(e->ws, sizeof(double) * (size_t)(2 * n + 1) * n * batch);
    e->sH_cap = nn * batch;
    alloc(e->sH, sizeof(double) * e->sH_cap); alloc(e->sv, sizeof(double) * n * batch); alloc(e->sr, sizeof(double) * batch);
    alloc(e->schi2, sizeof(double) * batch); alloc(e->simu, sizeof(double) * 7 * HV_EKF_MAX_PREDICT_SAMPLES * batch);
    alloc(e->sstatus, sizeof(int) * batch); alloc(e->sdrop, sizeof(int) * batch); alloc(e->sactive, batch);
    alloc(e->err_dev, sizeof(int));
    alloc(e->visit_counts, 4 * sizeof(int) * (Ekf::VISIT_SLOTS + 1)); alloc(e->visit_order, sizeof(int) * (size_t)Ekf::VISIT_SLOTS * batch);
    alloc(e->visit_long, sizeof(int) * (size_t)Ekf::VISIT_SLOTS * batch); alloc(e->visit_long_count, sizeof(int) * Ekf::VISIT_SLOTS); alloc(e->visit_lists, 3 * sizeof(int) * (size_t)batch);
    if (ok && hipMemset(e->err_dev, 0, sizeof(int)) != hipSuccess) ok = false;
    if (!ok) { hv_ekf_destroy(h); return HV_ERR_NOMEM; }

    // initial state and covariance: EKFImplementation ctor, ekf.cpp:153-296
    std::vector<double> m0(n, 0.0), P0(nn, 0.0), Q0(144, 0.0);
    auto sq = [](double x) { return x * x; };
    m0[hv::ORI] = 1.0; m0[hv::BAT] = m0[hv::BAT + 1] = m0[hv::BAT + 2] = 1.0;
    for (int i = 0; i < 3; i++) {
        P0[(size_t)(hv::POS + i) * n + hv::POS + i] = sq(par->noiseInitialPos);
        P0[(size_t)(hv::VEL + i) * n + hv::VEL + i] = sq(par->noiseInitialVel);
        P0[(size_t)(hv::BGA + i) * n + hv::BGA + i] = sq(par->noiseInitialBGA);
        P0[(size_t)(hv::BAA + i) * n + hv::BAA + i] = sq(par->noiseInitialBAA);
        P0[(size_t)(hv::BAT + i) * n + hv::BAT + i] = sq(par->noiseInitialBAT);
    }
    for (int i = 0; i < 4; i++) P0[(size_t)(hv::ORI + i) * n + hv::ORI + i] = 1.0;
    P0[(size_t)hv::SFT * n + hv::SFT] = sq(par->noiseInitialSFT);
    for (int cidx = 0; cidx < e->cam; cidx++) {
        const int b0 = hv::CAM + cidx * hv::POSE;
        for (int i = 0; i < 3; i++) P0[(size_t)(b0 + i) * n + b0 + i] = sq(par->noiseInitialPosTrail);
        for (int i = 3; i < 7; i++) P0[(size_t)(b0 + i) * n + b0 + i] = sq(par->noiseInitialOriTrail);
    }
    for (int i = 0; i < 3; i++) { Q0[(hv::Q_ACC + i) * 13] = sq(par->noiseProcessAcc); Q0[(hv::Q_GYRO + i) * 13] = sq(par->noiseProcessGyro); }
    for (auto &x : P0) x *= e->noise_scale;
    for (auto &x : Q0) x *= e->noise_scale;
    int rc = HV_OK;
    for (int b = 0; b < batch && rc == HV_OK; b++) {
        rc = hv_ekf_set_state(h, b, m0.data(), P0.data());
        if (rc == HV_OK) rc = hv_ekf_set_process_noise(h, b, Q0.data());
    }
    if (rc != HV_OK) { hv_ekf_destroy(h); return rc; }
    *out = h;
    return HV_OK;
}

void hv_vu_default_params(hv_vu_params *p)
{
    if (!p) return;
    *p = hv_vu_params{};
    p->triangulationConvergenceThreshold = 1e-2; p->triangulationConvergenceR = 11.0;            // parameter_definitions.c:37-44
    p->triangulationRcondThreshold = 1e-8; p->triangulationGaussNewtonIterations = 10;
    p->triangulationMinDist = 0; p->triangulationMaxDist = 1e300;
    p->estimateImuCameraTimeShift = 1;                                                          // :163
    p->trackRmseThreshold = -1.0; p->trackOutlierThresholdGrowthFactor = 1.0;                   // :21,27
    p->useStereo = 0;
    const double imu[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1};                                         // :178 imuToCameraMatrix (symmetric)
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        const double v = r < 3 && c < 3 ? imu[3 * r + c] : (r == c ? 1.0 : 0.0);
        p->imuToCamera[4 * r + c] = v; p->secondImuToCamera[4 * r + c] = v;
    }
    const double tr[3] = {0.0075, 0.013, -0.0003};                                               // :187 stereoCameraTranslation
    for (int r = 0; r < 3; ++r) p->secondImuToCamera[4 * r + 3] += tr[r];                         // tracker/util.cpp:100-105
}

static int vu_fill_args(Ekf *e, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                        const double *y, hv::VuPrepareArgs &a)
{
    if (!p || !idx || !feat || !vel || np < 2 || np > e->cam + 1) return HV_ERR_INVALID;
    a = hv::VuPrepareArgs{};
    a.batch = e->batch; a.n = e->n; a.np = np; a.stereo = p->useStereo ? 1 : 0;
    a.m = e->m; a.pose_index = idx; a.features = feat; a.velocities = vel; a.y = y;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) {
        a.imu_to_cam[0][4 * r + c] = p->imuToCamera[4 * r + c];
        a.imu_to_cam[1][4 * r + c] = p->secondImuToCamera[4 * r + c];
    }
    a.conv_threshold = p->triangulationConvergenceThreshold; a.conv_r = p->triangulationConvergenceR;
    a.rcond_threshold = p->triangulationRcondThreshold; a.min_dist = p->triangulationMinDist; a.max_dist = p->triangulationMaxDist;
    a.gn_iters = (int)p->triangulationGaussNewtonIterations; a.est_shift = p->estimateImuCameraTimeShift ? 1 : 0;
    a.defer_h = e->c->knob.ekf_defer_jacobian != 0;
    a.linear = p->useLinearTriangulation ? 1 : 0;
    // adaptive outlier thresholds (ABI 3): the RMSE test applies everywhere the fused gate runs; the per-filter growth only inside a
    // frame loop (visual_frame_dev_impl hands the multiplier array over through Ekf::gate_scale_on)
    a.rmse_thr = p->trackRmseThreshold; a.growth = p->trackOutlierThresholdGrowthFactor;
    a.gate_scale = e->gate_scale_on ? e->gate_scale : nullptr;
    if (!(a.growth > 0.0)) return HV_ERR_INVALID;
    return HV_OK;
}

int hv_ekf_visual_prepare_dev(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                              const double *y, double *H_dev, double *v_dev, double *f_dev, double *pf_dev, int *status_dev,
                              unsigned char *active_dev)
{
    if (!h || !H_dev || !v_dev || !pf_dev || !status_dev) return HV_ERR_INVALID;
    Ekf *e = &h->e;
    hv::VuPrepareArgs a;
    int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
    if (rc != HV_OK) return rc;
    a.H = H_dev; a.v = v_dev; a.f = f_dev; a.pf = pf_dev; a.status = status_dev; a.active = active_dev;
    return hv::launch_vu_prepare(e->c, a);
}

static int visual_track_dev_impl(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                                 const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev,
                                 double *chi2_dev, double *pf_dev, int *success_counter_dev, int max_successful,
                                 const int *np_rec_dev = nullptr);

// shapes of a track visit: the longest track the short class's fused two-per-CU kernels serve, whether tracks of np poses take the
// long-class launches (49 .. 96 rows), and whether a ragged visit of up to np poses runs as TWO length classes
struct VisitShape { int ncam, np_short, rows; bool long_ok, two_class; };
// (r06: where the split form serves the visit -- stereo, iterative triangulation, more filters than CUs -- the short class ends at
//  12 poses = 48 rows, what the record-fed gate stages three to a CU, instead of 11: 4 % of a ragged visit's records leave the long
//  class, its 158 KB gate and its two block updates)
static VisitShape visit_shape(const Ekf *e, int np, bool stereo, bool linear)
{
    VisitShape v{};
    v.ncam = stereo ? 2 : 1; v.np_short = 22 / v.ncam; v.rows = 2 * np * v.ncam;
    if (hv::vu_split_short_ok(e->c, e->n, stereo, e->batch, linear)) v.np_short = hv::vu_split_short_np(e->c);
    v.long_ok = e->c->knob.ekf_fused_gate != 0 && e->n <= 160 && v.rows > 48 && v.rows <= 96 && v.rows < HV_CHI2INV95_N && (v.rows + 3) / 4 * 2 <= 48;
    v.two_class = np > v.np_short && v.long_ok && hv::vu_fused_supported(e->c, e->n, v.np_short, stereo, e->batch);
    return v;
}

int hv_ekf_visual_track_dev(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                            const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev,
                            double *chi2_dev, double *pf_dev)
{
    return visual_track_dev_impl(h, p, np, idx, feat, vel, y, r_gate, r_update, status_dev, gate_status_dev, chi2_dev, pf_dev, nullptr, 0);
}

int hv_ekf_visual_track_limited_dev(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                                    const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev,
                                    double *chi2_dev, double *pf_dev, int *success_counter_dev, int max_successful)
{
    if (!success_counter_dev) return HV_ERR_INVALID;
    // maxSuccessfulVisualUpdates <= 0 is the reference's "no limit" (backend.cpp:1233), as in the frame entry points (r03 advisor)
    if (max_successful <= 0) max_successful = INT_MAX;
    return visual_track_dev_impl(h, p, np, idx, feat, vel, y, r_gate, r_update, status_dev, gate_status_dev, chi2_dev, pf_dev,
                                 success_counter_dev, max_successful);
}

// One track visit of a session with a hybrid map (odometry.hybridMapSize > 0; backend.cpp:1016, 1075-1082, 1146, 1160-1168):
//   map_update_dev [batch]: the map point a mapPointUpdate track belongs to (>= 0) -- the point is read from the state, status
//                           HV_TRI_HYBRID, H carries dip R in the point's columns -- or -1 for a pose-trail track;
//   map_offer_dev [batch]:  the slot ekfStateIndex.offerMapPoint would hand to this track if the gate accepts it (-1: none; the offer
//                           does not depend on the filter, so the adapter evaluates it up front): such an inlier is INSERTED as a map
//                           point instead of being applied.
// Dense kernels (the state is wider than 160 columns). Either array may be NULL.
int hv_ekf_visual_track_hybrid_dev(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                                   const double *y, const int *map_update_dev, const int *map_offer_dev, double r_gate, double r_update,
                                   int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev)
{
    if (!h || !status_dev || !gate_status_dev || !y) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (e->map_dim <= 0 && (map_update_dev || map_offer_dev)) return HV_ERR_INVALID;
    hv::VuPrepareArgs a;
    int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
    if (rc != HV_OK) return rc;
    if (a.rmse_thr >= 0.0) return HV_ERR_UNSUPPORTED;                  // (the dense gate has no RMSE test)
    const int rows = 2 * np * (a.stereo ? 2 : 1);
    if (rows > e->max_rows || rows >= HV_CHI2INV95_N) return HV_ERR_INVALID;
    if (e->vu_rows < rows) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        void *old[] = {e->vuH, e->vuv};
        for (void *q : old) if (q) (void)hipFree(q);
        e->vuH = e->vuv = nullptr; e->vu_rows = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuH), sizeof(double) * (size_t)rows * e->n * e->batch));
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuv), sizeof(double) * (size_t)rows * e->batch));
        if (!e->vupf) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vupf), sizeof(double) * 3 * e->batch));
        if (!e->vuactive) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuactive), e->batch));
        if (!e->vuacol) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuacol), sizeof(int) * (size_t)e->n * e->batch));
        e->vu_rows = rows;
    }
    const double ns = e->noise_scale;
    double *pf = pf_dev ? pf_dev : e->vupf;
    a.H = e->vuH; a.v = e->vuv; a.f = nullptr; a.pf = pf; a.status = status_dev; a.active = e->vuactive; a.gate_status = gate_status_dev;
    a.map_index = map_update_dev; a.map_base = e->n - e->map_dim;
    rc = hv::launch_vu_prepare(c, a);
    if (rc != HV_OK) return rc;
    // visualTrackOutlierCheck, then -- per filter -- insertMapPoint or updateVisualTrack where it passed
    rc = hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_gate * r_gate * ns, 0, 0, 0, chi2_dev, gate_status_dev, e->vuactive);
    if (rc != HV_OK) return rc;
    hipLaunchKernelGGL(hv::ekf_hybrid_insert_kernel, dim3(e->batch), dim3(256), 0, c->stream, e->n, e->n - e->map_dim, e->m, e->P, e->vuactive,
                       gate_status_dev, map_update_dev, map_offer_dev, pf, e->sactive);
    HV_HIP(c, hipGetLastError());
    return hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr, nullptr, e->sactive, gate_status_dev);
}

static int visual_track_dev_impl(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                                 const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev,
                                 double *chi2_dev, double *pf_dev, int *success_counter_dev, int max_successful,
                                 const int *np_rec_dev)
{
    if (!h || !status_dev || !gate_status_dev || !y) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hv::VuPrepareArgs a;
    int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
    if (rc != HV_OK) return rc;
    const int rows = 2 * np * (a.stereo ? 2 : 1);
    if (rows > e->max_rows) return HV_ERR_INVALID;
    if (e->vu_rows < rows) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        void *old[] = {e->vuH, e->vuv};
        for (void *q : old) if (q) (void)hipFree(q);
        e->vuH = e->vuv = nullptr; e->vu_rows = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuH), sizeof(double) * (size_t)rows * e->n * e->batch));
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuv), sizeof(double) * (size_t)rows * e->batch));
        if (!e->vupf) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vupf), sizeof(double) * 3 * e->batch));
        if (!e->vuactive) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuactive), e->batch));
        if (!e->vuacol) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vuacol), sizeof(int) * (size_t)e->n * e->batch));
        e->vu_rows = rows;
    }
    // ragged batch (filters with tracks of different lengths, or none, in one visit): np is the longest track = the record stride;
    // the prepare launch writes every filter's row count for the gate / update launch
    const int *nr_rec = nullptr;
    if (np_rec_dev) {
        if (!e->vurows) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vurows), sizeof(int) * e->batch));
        a.np_rec = np_rec_dev; a.rows_out = e->vurows; nr_rec = e->vurows;
    }
    a.H = e->vuH; a.v = e->vuv; a.f = nullptr; a.pf = pf_dev ? pf_dev : e->vupf; a.status = status_dev; a.active = e->vuactive;
    a.gate_status = gate_status_dev;                       // preset to NOT_COMPUTED; the gate overwrites it where it runs
    a.success_counter = success_counter_dev; a.max_successful = max_successful;
    const double ns = e->noise_scale;
    hipStream_t main_stream = c->stream;
    // r03 default: visualTrackOutlierCheck runs INSIDE the prepare launch on the active columns of H (vu_gate kernels: the Jacobian
    // of a rejected track never leaves LDS and only P(a, a) is read); updateVisualTrack then runs where the gate said INLIER, staging the
    // compact Jacobian through its column map (7 of 20 visits at most -- backend.cpp:1233-1238 -- pay the full H P + downdate).
    // Long tracks (more than 48 rows / 22 camera poses: 12 .. 21 stereo poses): prepare + column-sparse gate of up to 96 rows in ONE
    // launch (r04: vu_gate_long_kernel; r03: vu_compact_kernel + ekf_sparse_gate_big_kernel, kept behind knob ekf_long_fused = 0) and the
    // update as TWO block updates of at most 48 rows each on the P-resident kernel (UpdateArgs::half).
    const VisitShape shape = visit_shape(e, np, a.stereo != 0, a.linear != 0);
    const int ncam = shape.ncam, np_short = shape.np_short;
    const bool long_ok = shape.long_ok;
    // compaction lists of this visit (VuPrepareArgs): zeroed here, filled by the kernels, consumed by the launches behind them
    const bool own_counts = e->visit_slot < 0 || e->visit_slot >= Ekf::VISIT_SLOTS;
    int *counts = e->visit_counts + 4 * (own_counts ? Ekf::VISIT_SLOTS : e->visit_slot);
    int *cnt_inl = counts, *cnt_long = counts + 1, *cnt_inl_long = counts + 2;
    int *list_inl = e->visit_lists, *list_long = e->visit_lists + e->batch, *list_inl_long = e->visit_lists + 2 * (size_t)e->batch;
    // frame loop with sorted visits (launch_visit_order): the long class's records are known -- longest first -- before the fused launch runs
    const bool presorted = !own_counts && e->visit_order_ok;
    if (presorted) { cnt_long = e->visit_long_count + e->visit_slot; list_long = e->visit_long + (size_t)e->visit_slot * e->batch; }
    if (own_counts) HV_HIP(c, hipMemsetAsync(counts, 0, 4 * sizeof(int), main_stream));
    // short_upd (ragged two-class visits): issues the short class's update, or only prepares it (non-null argument) so that it shares
    // a grid with the first block update of the long class
    using ShortUpd = std::function<int(hv::UpdateLaunch *)>;
    // buffers of the long class: compact Jacobians, residuals, column lists, active flags, block 1's mean step; the fork / join events
    auto ensure_long = [&]() -> int {
        if (!e->ev_fork) {
            HV_HIP(c, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
            HV_HIP(c, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        }
        if (e->side_rows < rows) {
            HV_HIP(c, hipStreamSynchronize(main_stream));
            HV_HIP(c, hipStreamSynchronize(c->aux_stream));
            if (e->sideH) (void)hipFree(e->sideH);
            if (e->sidev) (void)hipFree(e->sidev);
            e->sideH = e->sidev = nullptr; e->side_rows = 0;
            HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->sideH), sizeof(double) * (size_t)rows * e->n * e->batch));
            HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->sidev), sizeof(double) * (size_t)rows * e->batch));
            if (!e->side_active) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->side_active), e->batch));
            if (!e->side_acol) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->side_acol), sizeof(int) * (size_t)e->n * e->batch));
            if (!e->side_dm) {
                HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->side_dm), sizeof(double) * (size_t)e->n * e->batch));
                HV_HIP(c, hipMemsetAsync(e->side_dm, 0, sizeof(double) * (size_t)e->n * e->batch, main_stream));   // (r03 advisor: never read uninitialised)
            }
            e->side_rows = rows;
        }
        return HV_OK;
    };
    // factor records of the split form (knob ekf_split_tri, r06): vu_tri_kernel -> record -> record-fed gate
    auto ensure_tri = [&]() -> int {
        const int stride = hv::vu_tri_rec_stride(np, ncam);
        if (e->tri_stride >= stride) return HV_OK;
        HV_HIP(c, hipStreamSynchronize(main_stream));
        if (c->aux_stream) HV_HIP(c, hipStreamSynchronize(c->aux_stream));
        if (e->tri_rec) (void)hipFree(e->tri_rec);
        e->tri_rec = nullptr; e->tri_stride = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->tri_rec), sizeof(double) * (size_t)stride * e->batch));
        e->tri_stride = stride;
        return HV_OK;
    };
    // the gate launch `g` of one length class in split form: the triangulation front first, on the same stream; false = not a shape
    // the split form serves (the caller issues the fused launch)
    // phase: 0 = both launches, 1 = the triangulation only, 2 = the gate only (one-stream visits interleave the two classes' launches:
    // knob ekf_long_first)
    auto split_launch = [&](hv::VuPrepareArgs &g, hipStream_t stream, int *rc_out, int phase = 0) -> bool {
        if (!hv::vu_split_supported(c, g, g.fused)) return false;
        int rc2 = ensure_tri();
        g.tri_rec = e->tri_rec; g.tri_stride = e->tri_stride;
        if (rc2 == HV_OK && phase != 2) rc2 = hv::launch_vu_tri(c, g, stream);
        if (rc2 == HV_OK && phase != 1) { g.from_rec = 1; rc2 = hv::launch_vu_prepare(c, g, stream); }
        *rc_out = rc2;
        return true;
    };
    // prepare + gate of the long class on `stream` (nothing else touches c->stream: r03 swapped the context's stream for these calls)
    auto long_prepare_gate = [&](hv::VuPrepareArgs l_, double *Hc, double *vv, int *acol, unsigned char *act, bool listed, hipStream_t stream, int phase = 0) -> int {
        l_.H = nullptr; l_.Hc = Hc; l_.v = vv; l_.acol = acol; l_.na_max = 7 * np + 1; l_.active = act; l_.chi2 = chi2_dev;
        if (listed) { l_.rec_count = cnt_long; l_.rec_list = list_long; }
        const bool adaptive_gate = l_.rmse_thr >= 0.0 || l_.gate_scale;     // (served by the fused gates only)
        if (c->knob.ekf_long_fused == 0 && adaptive_gate) return HV_ERR_UNSUPPORTED;
        if (c->knob.ekf_long_fused != 0) {
            l_.fused = 3; l_.P = e->P; l_.rd_gate = r_gate * r_gate * ns; l_.noise_scale = ns;
            l_.inl_count = cnt_inl_long; l_.inl_list = list_inl_long;
            int rc_s = HV_OK;
            if (split_launch(l_, stream, &rc_s, phase)) return rc_s;
            if (phase == 1) return HV_OK;                      // (not a split shape: the fused launch is the gate phase)
            return hv::launch_vu_prepare(c, l_, stream);
        }
        l_.fused = 2;
        const int rc2 = hv::launch_vu_prepare(c, l_, stream);
        if (rc2 != HV_OK) return rc2;
        return hv::ekf_launch_sparse_gate(e, np, ncam, Hc, vv, acol, nr_rec, act, r_gate * r_gate * ns, chi2_dev, gate_status_dev,
                                          listed ? cnt_long : nullptr, listed ? list_long : nullptr, cnt_inl_long, list_inl_long, stream);
    };
    // the two block updates of the long class's inliers on the context stream; with short_upd the short class's update shares their grids
    auto long_updates = [&](double *Hc, double *vv, int *acol, unsigned char *act, double *dm, const ShortUpd *short_upd) -> int {
        const int half_rows = 2 * ((rows + 3) / 4);            // the longer of the two blocks of the longest record
        const int na_max = 7 * np + 1;
        hv::CompactH h1{acol, na_max, ncam, 1, rows, dm, cnt_inl_long, list_inl_long, gate_status_dev}, h2{acol, na_max, ncam, 2, rows, dm, cnt_inl_long, list_inl_long};
        // (block 1 may turn a record's gate status into CHI2 when it meets a non-positive pivot: block 2 then skips it -- r03 advisor)
        auto block1 = [&](hv::UpdateLaunch *defer) -> int {
            return hv::ekf_launch_update(e, half_rows, e->n, Hc, vv, nullptr, r_update * r_update * ns, 1, 0, -1, nullptr, nullptr, act, gate_status_dev,
                                         nullptr, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nr_rec, &h1, rows, defer);
        };
        auto block2 = [&](hv::UpdateLaunch *defer) -> int {
            return hv::ekf_launch_update(e, half_rows, e->n, Hc, vv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr, nullptr, act, gate_status_dev,
                                         success_counter_dev, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nr_rec, &h2, rows, defer);
        };
        int rc2 = HV_OK;
        if (short_upd) {
            // (short class beside block 1, what did not fit on the chip beside block 2: see ekf_update_dual_kernel)
            hv::UpdateLaunch us, u1, u2;
            bool dual = false;
            rc2 = (*short_upd)(&us);
            if (rc2 == HV_OK) rc2 = block1(&u1);
            if (rc2 == HV_OK) rc2 = block2(&u2);
            if (rc2 == HV_OK) rc2 = hv::ekf_launch_update_dual(e, us, u1, 0, &dual);
            if (rc2 == HV_OK && dual) return hv::ekf_launch_update_dual(e, us, u2, 1, &dual);      // (same shapes: accepted again)
            if (rc2 == HV_OK) { rc2 = (*short_upd)(nullptr); if (rc2 == HV_OK) rc2 = block1(nullptr); }
        } else rc2 = block1(nullptr);
        if (rc2 != HV_OK) return rc2;
        return block2(nullptr);
    };
    // Ragged batch with long AND short tracks: two length CLASSES per visit. The short tracks -- 4 of 5 at the reference's defaults, see
    // bench.py sample_track_lengths -- run on the fused two-per-CU kernels, the long ones through the launches above; the records of a visit
    // belong to different filters, so the two launch sequences are independent up to the shared update grids. Every launch skips the
    // other class's records (VuPrepareArgs::np_lo / np_hi, `active`). Schedule of a visit inside a frame loop over more filters than CUs
    // (knob ekf_side_stream != 0, default; HIP-graph capturable: fork at the start of the visit, join in front of the updates):
    //     second stream (Ctx::aux_stream):  prepare + gate (long)              -- enqueued FIRST: it takes its CUs while all are free
    //     context stream:                   fused prepare + gate (short)  | join |  short update + long block 1  ->  rest of short + long block 2
    // Stand-alone visits and small batches: everything on the context's stream, short class first (its launch collects the long list).
    // (r03 also had the whole long chain on the second stream -- knob values 1 and 4 -- and the long class enqueued BEHIND the fused
    //  launch -- 2: all measured slower, removed in r04 together with the stream swap they needed.)
    if (np_rec_dev && shape.two_class) {
        rc = ensure_long();
        if (rc != HV_OK) return rc;
        bool use_aux = c->knob.ekf_side_stream != 0 && presorted && c->aux_stream;
        if (use_aux && c->knob.ekf_side_stream == 5) {
            // lanes (knob value 5, their default): no fork while the context stream is being CAPTURED. A captured fork does not run on
            // the library's high-priority second stream when the graph is replayed but on a stream the graph instance creates for the
            // branch -- default priority, bound to whichever hardware queue the process history left least used --, which is exactly the
            // placement lottery the lanes exist to end (r04, scripts/lanes_probe.py: 2 x 1024 sequences 19.2 / 16.3 ms per step with the
            // captured fork after two different process histories, 15.84 ms without it)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(main_stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) use_aux = false;
        }
        hv::VuPrepareArgs s_ = a;                              // class "short": 2 .. np_short poses (and the records without a track)
        s_.np_lo = 2; s_.np_hi = np_short; s_.class_inactive = 1;
        s_.fused = 1; s_.H = nullptr; s_.Hc = e->vuH; s_.acol = e->vuacol; s_.na_max = 7 * np + 1; s_.P = e->P;
        s_.rd_gate = r_gate * r_gate * ns; s_.noise_scale = ns; s_.chi2 = chi2_dev;
        s_.inl_count = cnt_inl; s_.inl_list = list_inl;
        if (!presorted) { s_.long_count = cnt_long; s_.long_list = list_long; }      // (else nothing to collect)
        if (presorted) s_.order = e->visit_order + (size_t)e->visit_slot * e->batch;     // (frame loop: sorted once per frame)
        hv::VuPrepareArgs l_ = a;
        l_.np_lo = np_short + 1; l_.np_hi = np; l_.class_inactive = 1;
        // fork ... join: every exit between the two goes through the join (a HIP-graph capture must not be left with a dangling fork)
        bool forked = false;
        if (use_aux) {
            hipError_t he = hipEventRecord(e->ev_fork, main_stream);
            if (he == hipSuccess) he = hipStreamWaitEvent(c->aux_stream, e->ev_fork, 0);
            if (he != hipSuccess) return hv::hip_fail(c, he, "fork onto the second stream");
            forked = true;
        }
        rc = HV_OK;
        // which class runs on which stream when the visit forks: the launch that has to START first -- the long class's, which needs
        // whole CUs and finds them only while the chip is empty -- belongs on the stream that does NOT wait for the fork event.
        // r04 default (knob 6, and 5 = the lanes' form of it): long class on the context stream, short class on the second stream.
        // Knob 3 = r03's arrangement: the long class on the second stream, enqueued first -- but released by the fork event about when
        // the short class's 810 two-per-CU workgroups are, so that part of it got its CUs a round late (144 us under that load against
        // 95 alone, profiles/r04/kernel_stats.csv). Measured, one context of 1024 sequences: eager 10.13 -> 9.82 ms per step, HIP-graph
        // replay 9.67 -> 9.63 (profiles/r04/lanes_probe.txt, session 9).
        const bool swap = forked && c->knob.ekf_side_stream != 3;
        hipStream_t long_stream = forked && !swap ? c->aux_stream : main_stream, short_stream = swap ? c->aux_stream : main_stream;
        const bool long_first = presorted && (forked || c->knob.ekf_long_first == 1);
        auto short_phase = [&](int phase) -> int {
            int rc_s = HV_OK;
            if (split_launch(s_, short_stream, &rc_s, phase)) return rc_s;
            return phase == 1 ? HV_OK : hv::launch_vu_prepare(c, s_, short_stream);
        };
        // one-stream sorted visits, knob ekf_long_first: 0 = short class (triangulation, gate) then long class; 1 = long class first;
        // 2 .. 4 (split form, r06) = both triangulations in front of both gates: T long, T short, G long, G short / T short, T long,
        // G short, G long / T long, T short, G short, G long (profiles/r06/visit_launch_order_sweep.txt)
        const int order = presorted && !forked ? c->knob.ekf_long_first : -1;
        if (order >= 2 && order <= 4) {
            const bool lf = order != 3;
            if (lf) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, main_stream, 1);
            if (rc == HV_OK) rc = short_phase(1);
            if (rc == HV_OK && !lf) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, main_stream, 1);
            const bool gl_first = order == 2;
            if (rc == HV_OK && gl_first) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, main_stream, 2);
            if (rc == HV_OK) rc = short_phase(2);
            if (rc == HV_OK && !gl_first) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, main_stream, 2);
        } else {
        if (long_first) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, long_stream);
        if (rc == HV_OK) rc = short_phase(0);
        if (rc == HV_OK && !long_first) rc = long_prepare_gate(l_, e->sideH, e->sidev, e->side_acol, e->side_active, true, main_stream);
        }
        if (forked) {
            hipError_t he = hipEventRecord(e->ev_join, c->aux_stream);
            if (he == hipSuccess) he = hipStreamWaitEvent(main_stream, e->ev_join, 0);
            if (he != hipSuccess && rc == HV_OK) rc = hv::hip_fail(c, he, "join of the second stream");
        }
        if (rc != HV_OK) return rc;
        const hv::CompactH ch{e->vuacol, s_.na_max, ncam, 0, 0, nullptr, cnt_inl, list_inl};
        const ShortUpd short_upd = [&](hv::UpdateLaunch *defer) -> int {
            return hv::ekf_launch_update(e, 2 * np_short * ncam, e->n, e->vuH, e->vuv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr, nullptr,
                                         e->vuactive, gate_status_dev, success_counter_dev, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                                         nullptr, 0, nr_rec, &ch, rows, defer);
        };
        // knob ekf_dual_update (default 1): the short class's update shares a grid with the first block update of the long class (the two
        // serve different filters)
        const bool pair = c->knob.ekf_dual_update != 0;
        if (!pair) { rc = short_upd(nullptr); if (rc != HV_OK) return rc; }
        return long_updates(e->sideH, e->sidev, e->side_acol, e->side_active, e->side_dm, pair ? &short_upd : nullptr);
    }
    if (long_ok && np > np_short) {                            // every record of the launch may be long (uniform 12 .. 21 stereo poses, or ragged)
        rc = ensure_long();
        if (rc != HV_OK) return rc;
        rc = long_prepare_gate(a, e->vuH, e->vuv, e->vuacol, e->vuactive, false, main_stream);
        if (rc != HV_OK) return rc;
        return long_updates(e->vuH, e->vuv, e->vuacol, e->vuactive, e->side_dm, nullptr);
    }
    if (hv::vu_fused_supported(c, e->n, np, a.stereo, e->batch)) {
        // knob ekf_fused_gate: -1 auto / 1 = the gate inside the prepare launch (vu_gate kernels); 2 = its own launch (vu_compact kernels +
        // ekf_sparse_gate_kernel, three 43 KB workgroups per CU). Measured at 1024 filters x 10 stereo poses (r03, scripts/vu_microbench.py):
        // 156 us against 84 + 67 us per visit, and 9.0 against 9.25 ms for the chained C3 step -- the gate costs ~17 us of a CU per filter
        // in either form (a 40-pivot Cholesky chain on one wave + 430 MFMAs at low occupancy), so the form with one launch and no round trip
        // of Hc through HBM stays the default at every batch size.
        const int fg = c->knob.ekf_fused_gate;
        const bool split_gate = fg == 2;
        if (split_gate && (a.rmse_thr >= 0.0 || a.gate_scale)) return HV_ERR_UNSUPPORTED;   // (adaptive thresholds: fused gates only)
        a.fused = split_gate ? 2 : 1; a.H = nullptr; a.Hc = e->vuH; a.acol = e->vuacol; a.na_max = 7 * np + 1; a.P = e->P;
        a.rd_gate = r_gate * r_gate * ns; a.noise_scale = ns; a.chi2 = chi2_dev;
        if (!split_gate) { a.inl_count = cnt_inl; a.inl_list = list_inl; }
        { int rc_s = HV_OK; rc = (!split_gate && split_launch(a, main_stream, &rc_s)) ? rc_s : hv::launch_vu_prepare(c, a); }
        if (rc != HV_OK) return rc;
        if (split_gate) {
            rc = hv::ekf_launch_sparse_gate(e, np, a.stereo ? 2 : 1, e->vuH, e->vuv, e->vuacol, nr_rec, e->vuactive, a.rd_gate, chi2_dev, gate_status_dev,
                                            nullptr, nullptr, cnt_inl, list_inl);
            if (rc != HV_OK) return rc;
        }
        const hv::CompactH ch{e->vuacol, a.na_max, a.stereo ? 2 : 1, 0, 0, nullptr, cnt_inl, list_inl};
        return hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr, nullptr,
                                     e->vuactive, gate_status_dev, success_counter_dev, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                                     nullptr, 0, nr_rec, &ch);
    }
    // (the dense kernels below know neither the RMSE test nor the per-filter threshold growth)
    if (a.rmse_thr >= 0.0 || a.gate_scale) return HV_ERR_UNSUPPORTED;
    rc = hv::launch_vu_prepare(c, a);
    if (rc != HV_OK) return rc;
    // Dense path (tracks of more than 48 rows, filters wider than 160, knob ekf_fused_gate = 0): visualTrackOutlierCheck with chiOutlierR,
    // then updateVisualTrack with visualR where everything passed: one launch when the shape runs on the register-resident kernel
    // (mode 3), otherwise a gate launch and an update launch.
    // knob ekf_stream_gate = 1 (experiment): the streaming gate kernel (two filters per CU) for everybody, then the
    // register-resident update where the gate passed. Measured at 1024 filters (r02): a rejected track costs 0.110 ms instead of the
    // fused launch's 0.130, an accepted one 0.110 + 0.21 instead of 0.23.
    if (c->knob.ekf_stream_gate == 1 && !nr_rec) {
        bool done = false;
        rc = hv::ekf_launch_gate_stream(e, rows, e->n, e->vuH, e->vuv, r_gate * r_gate * ns, chi2_dev, gate_status_dev, e->vuactive,
                                        success_counter_dev, max_successful, &done);
        if (rc != HV_OK) return rc;
        if (done)
            return hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr,
                                         nullptr, e->vuactive, gate_status_dev, success_counter_dev);
    }
    bool fused = false;
    rc = hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_gate * r_gate * ns, 3, 0, 1, chi2_dev,
                               gate_status_dev, e->vuactive, nullptr, success_counter_dev, r_update * r_update * ns, &fused,
                               0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nr_rec);
    if (rc != HV_OK || fused) return rc;
    rc = hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_gate * r_gate * ns, 0, 0, 0, chi2_dev,
                               gate_status_dev, e->vuactive, nullptr, nullptr, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nr_rec);
    if (rc != HV_OK) return rc;
    return hv::ekf_launch_update(e, rows, e->n, e->vuH, e->vuv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr,
                                 nullptr, e->vuactive, gate_status_dev, success_counter_dev, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr,
                                 nullptr, 0, nr_rec);
}


// work buffers of the speculative frame loops and the batch loop: one record per (track, filter) -- compact or dense Jacobian, residual,
// point, flags, column list, rows -- plus the per-filter cursors; grown on first use of a shape and never shrunk in either dimension
// (r04 advisor: reallocating to exactly (rec, rows) let alternating shapes -- a short and a long frame, the speculative and the batch
// loop -- free and allocate on every call, which a stream capture cannot hold)
static int ensure_spec_buffers(Ekf *e, size_t rec, int rows)
{
    Ctx *c = e->c;
    if (e->sp_records >= rec && e->sp_rows >= rows) return HV_OK;
    rec = std::max(rec, e->sp_records); rows = std::max(rows, e->sp_rows);
    const size_t B = (size_t)e->batch;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    void *old[] = {e->spH, e->spv, e->sppf, e->spactive, e->spcursor, e->spepoch, e->spcursor2, e->sppub, e->spacol, e->sprows};
    for (void *q : old) if (q) (void)hipFree(q);
    e->spH = e->spv = e->sppf = nullptr; e->spactive = nullptr; e->spcursor = e->spepoch = e->spcursor2 = e->sppub = e->spacol = e->sprows = nullptr; e->sp_records = 0;
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spH), sizeof(double) * rec * rows * e->n));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spv), sizeof(double) * rec * rows));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->sppf), sizeof(double) * rec * 3));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spactive), rec));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spcursor), sizeof(int) * B));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spepoch), sizeof(int) * rec));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spcursor2), sizeof(int) * B));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->sppub), sizeof(int) * rec));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->spacol), sizeof(int) * rec * e->n));
    HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->sprows), sizeof(int) * rec));
    e->sp_records = rec; e->sp_rows = rows;
    return HV_OK;
}

static int visual_frame_dev_impl(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np, const int *np_rec_dev, const int *idx,
                                 const double *feat, const double *vel,
                            const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev,
                            double *pf_dev, int *success_counter_dev, int max_successful)
{
    if (!h || n_tracks < 0 || !success_counter_dev) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    // maxSuccessfulVisualUpdates <= 0 is the reference's "no limit" (backend.cpp:1233: the test is `> 0 && count >= max`): every track
    // is visited, which a quota of n_tracks expresses exactly (r02 advisor)
    if (max_successful <= 0 || max_successful > n_tracks) max_successful = n_tracks > 0 ? n_tracks : 1;
    const size_t B = (size_t)e->batch, nt = (size_t)np * (p && p->useStereo ? 2 : 1);
    HV_HIP(c, hipMemsetAsync(success_counter_dev, 0, sizeof(int) * B, c->stream));          // updateSuccessCount = 0 (backend.cpp:1017)
    // adaptive outlier thresholds (backend.cpp:994-996,1192-1193): every filter starts the frame at the base thresholds; a rejected
    // track multiplies its filter's thresholds for the tracks behind it. Kept per filter on the device, reset here.
    const bool adaptive = p && p->trackOutlierThresholdGrowthFactor != 1.0;
    struct ScaleScope { Ekf *e; ~ScaleScope() { e->gate_scale_on = false; } } scale_scope{e};
    if (adaptive) {
        if (!e->gate_scale) HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->gate_scale), sizeof(double) * B));
        hipLaunchKernelGGL(hv::fill_doubles_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, c->stream, e->gate_scale, (int)B, 1.0);
        HV_HIP(c, hipGetLastError());
        e->gate_scale_on = true;
    }
    // Few sequences (one, for the reference's `main`): the frame is latency bound -- 20 dependent visits of a ~26 us prepare and a
    // ~34 us gate. While the GPU has idle CUs the loop is run SPECULATIVELY instead: a pass prepares and gates
    // EVERY pending track of a filter against the current (m, P) in parallel, applies the first inlier in visit order, and only the
    // tracks behind it are re-examined: <= min(max_successful, n_tracks) + 1 passes, the same statuses and the same filter as the
    // sequential loop (tracks in front of the first inlier saw the state they would have seen anyway).
    const int rows = 2 * (int)nt;
    // Long tracks (49 .. 84 rows; ragged frames whose longest track is that long: the reference's default stereo configuration, SURVEY
    // app. B) -- r04, VERDICT r03 item 6: the same speculative loop with the long build of the fused prepare + gate launch serving EVERY
    // pending record of whatever length (grid (filters, tracks), one 158 KB workgroup per CU while the chip is idle), and the first
    // pending inlier applied by the two block-update launches of the long class: a record of at most 48 rows whole by the first of them,
    // a longer one block by block (UpdateArgs::half_auto; sel_io hands the chosen track from the first launch to the second).
    // <= quota + 1 passes of three launches instead of n_tracks visits of four.
    if (!c->knob.ekf_no_speculation && !adaptive && n_tracks >= 2 && B * (size_t)n_tracks <= (size_t)c->num_cus && rows > 48 && p && idx && feat && vel && y &&
        c->knob.ekf_spec_split == 0 && c->knob.ekf_spec_mode != 3 && c->knob.ekf_long_fused != 0 && visit_shape(e, np, p->useStereo != 0, p->useLinearTriangulation != 0).long_ok) {
        const size_t rec = B * (size_t)n_tracks;
        { const int rc_sp = ensure_spec_buffers(e, rec, rows); if (rc_sp != HV_OK) return rc_sp; }
        if (!e->side_dm) {
            HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->side_dm), sizeof(double) * (size_t)e->n * B));
            HV_HIP(c, hipMemsetAsync(e->side_dm, 0, sizeof(double) * (size_t)e->n * B, c->stream));
        }
        HV_HIP(c, hipMemsetAsync(e->spcursor, 0, sizeof(int) * B, c->stream));
        HV_HIP(c, hipMemsetAsync(e->spcursor2, 0xFF, sizeof(int) * B, c->stream));                // sel_io: -1
        HV_HIP(c, hipMemsetAsync(e->spepoch, 0xFF, sizeof(int) * rec, c->stream));               // -1: nothing prepared yet
        hv::VuPrepareArgs a;
        int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
        if (rc != HV_OK) return rc;
        const double ns = e->noise_scale;
        const int ncam = a.stereo ? 2 : 1;
        a.H = nullptr; a.v = e->spv; a.f = nullptr; a.pf = pf_dev ? pf_dev : e->sppf; a.status = status_dev; a.active = e->spactive;
        a.gate_status = gate_status_dev; a.success_counter = success_counter_dev; a.max_successful = max_successful;
        a.spec_tracks = n_tracks; a.cursor = e->spcursor; a.epoch = e->spepoch;
        const int *nr_rec = nullptr;
        if (np_rec_dev) { a.np_rec = np_rec_dev; a.rows_out = e->sprows; nr_rec = e->sprows; }
        a.fused = 3; a.Hc = e->spH; a.acol = e->spacol; a.na_max = 7 * np + 1; a.P = e->P;
        a.rd_gate = r_gate * r_gate * ns; a.noise_scale = ns; a.chi2 = chi2_dev;
        hv::CompactH h1{e->spacol, a.na_max, ncam, 1, rows, e->side_dm, nullptr, nullptr, gate_status_dev}, h2{e->spacol, a.na_max, ncam, 2, rows, e->side_dm};
        h1.half_auto = h2.half_auto = 1; h1.sel_io = h2.sel_io = e->spcursor2; h2.epoch = e->spepoch;
        const int n_pass = (max_successful < n_tracks ? max_successful : n_tracks) + 1;
        for (int pass = 0; pass < n_pass; ++pass) {
            rc = hv::launch_vu_prepare(c, a);
            if (rc != HV_OK) return rc;
            for (const hv::CompactH *hh : {&h1, &h2}) {
                // (rows per launch: a whole short record of up to 48 rows, or the longer block -- 2 ceil(rows / 4) rows, the first camera's --
                //  of a long one; 48 for everything visit_shape admits today, rows <= 96)
                rc = hv::ekf_launch_update(e, std::max(48, 2 * ((rows + 3) / 4)), e->n, e->spH, e->spv, nullptr, r_update * r_update * ns, 1, 0, hh == &h1 ? -1 : 1, nullptr,
                                           nullptr, e->spactive, nullptr, success_counter_dev, 0.0, nullptr, 2, n_tracks, e->spcursor, max_successful,
                                           gate_status_dev, nullptr, nullptr, 0, nr_rec, hh, rows);
                if (rc != HV_OK) return rc;
            }
        }
        return HV_OK;
    }
    // (a growth factor != 1 makes the threshold of a track depend on the verdicts of the tracks in front of it: no parallel gating)
    if (!c->knob.ekf_no_speculation && !adaptive && n_tracks >= 2 && B * (size_t)n_tracks <= 256 && e->n <= 160 && rows <= 48 && p && idx && feat && vel && y) {
        const size_t rec = B * (size_t)n_tracks;
        { const int rc_sp = ensure_spec_buffers(e, rec, rows); if (rc_sp != HV_OK) return rc_sp; }
        HV_HIP(c, hipMemsetAsync(e->spcursor, 0, sizeof(int) * B, c->stream));
        HV_HIP(c, hipMemsetAsync(e->spcursor2, 0, sizeof(int) * B, c->stream));                   // (ping-pong partner: never read uninitialised)
        HV_HIP(c, hipMemsetAsync(e->spepoch, 0xFF, sizeof(int) * rec, c->stream));               // -1: nothing prepared yet
        hv::VuPrepareArgs a;
        int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
        if (rc != HV_OK) return rc;
        a.H = e->spH; a.v = e->spv; a.f = nullptr; a.pf = pf_dev ? pf_dev : e->sppf; a.status = status_dev; a.active = e->spactive;
        a.gate_status = gate_status_dev; a.success_counter = success_counter_dev; a.max_successful = max_successful;
        a.spec_tracks = n_tracks; a.cursor = e->spcursor; a.epoch = e->spepoch;
        const int *nr_rec = nullptr;                                  // ragged: per-record rows, written by the prepare launches
        if (np_rec_dev) { a.np_rec = np_rec_dev; a.rows_out = e->sprows; nr_rec = e->sprows; }
        const double ns = e->noise_scale;
        const int n_pass = (max_successful < n_tracks ? max_successful : n_tracks) + 1;
        // Pass forms (knob ekf_spec_mode; -1 = auto):
        //   2  (r03 default where the fused gate serves the shape) launch A: every pending track prepared AND gated on its active columns
        //      (vu_gate kernel, grid (filters, tracks)); launch B: per filter the first pending inlier is applied, the cursor moves behind it.
        //      No hand-shake between workgroups: what a pass computes does not depend on how the dispatcher places them.
        //   3  (r02) dense prepare, then ONE launch that gates every pending track and lets the first inlier apply itself after a spin-wait
        //      on the decisions in front of it. A wait that times out raises the filter batch's error word (hv_ekf_frame_error).
        //   knob ekf_spec_split = 1: dense prepare + dense gate-all + apply (the first r02 form).
        const int spec_mode = c->knob.ekf_spec_mode;
        const bool split = c->knob.ekf_spec_split != 0;
        if (!split && spec_mode != 3 && hv::vu_fused_supported(c, e->n, np, a.stereo, e->batch)) {
            a.fused = 1; a.H = nullptr; a.Hc = e->spH; a.acol = e->spacol; a.na_max = 7 * np + 1; a.P = e->P;
            a.rd_gate = r_gate * r_gate * ns; a.noise_scale = ns; a.chi2 = chi2_dev;
            const hv::CompactH ch{e->spacol, a.na_max, a.stereo ? 2 : 1};
            for (int pass = 0; pass < n_pass; ++pass) {
                rc = hv::launch_vu_prepare(c, a);
                if (rc != HV_OK) return rc;
                rc = hv::ekf_launch_update(e, rows, e->n, e->spH, e->spv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr,
                                           nullptr, e->spactive, nullptr, success_counter_dev, 0.0, nullptr, 2, n_tracks, e->spcursor, max_successful,
                                           gate_status_dev, nullptr, nullptr, 0, nr_rec, &ch);
                if (rc != HV_OK) return rc;
            }
            return HV_OK;
        }
        if (a.rmse_thr >= 0.0) return HV_ERR_UNSUPPORTED;             // (the dense pass forms have no RMSE test)
        if (!split) HV_HIP(c, hipMemsetAsync(e->sppub, 0, sizeof(int) * rec, c->stream));
        int *cur = e->spcursor, *nxt = e->spcursor2;
        for (int pass = 0; pass < n_pass; ++pass) {
            a.cursor = cur;
            rc = hv::launch_vu_prepare(c, a);
            if (rc != HV_OK) return rc;
            if (!split) {
                bool fused = false;
                rc = hv::ekf_launch_update(e, rows, e->n, e->spH, e->spv, nullptr, r_gate * r_gate * ns, 3, 0, 1, chi2_dev,
                                           gate_status_dev, e->spactive, nullptr, success_counter_dev, r_update * r_update * ns, &fused,
                                           3, n_tracks, cur, max_successful, nullptr, nxt, e->sppub, pass + 1, nr_rec);
                if (rc != HV_OK) return rc;
                if (fused) { int *sw = cur; cur = nxt; nxt = sw; continue; }
                // (mode 3 not available for this shape: the two launches below, on the same cursor)
            }
            rc = hv::ekf_launch_update(e, rows, e->n, e->spH, e->spv, nullptr, r_gate * r_gate * ns, 0, 0, 0, chi2_dev,
                                       gate_status_dev, e->spactive, nullptr, success_counter_dev, 0.0, nullptr, 1, n_tracks, cur, max_successful,
                                       nullptr, nullptr, nullptr, 0, nr_rec);
            if (rc != HV_OK) return rc;
            rc = hv::ekf_launch_update(e, rows, e->n, e->spH, e->spv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr,
                                       nullptr, e->spactive, nullptr, success_counter_dev, 0.0, nullptr, 2, n_tracks, cur, max_successful,
                                       gate_status_dev, nullptr, nullptr, 0, nr_rec);
            if (rc != HV_OK) return rc;
        }
        return HV_OK;
    }
    HV_HIP(c, hipMemsetAsync(e->visit_counts, 0, 4 * sizeof(int) * Ekf::VISIT_SLOTS, c->stream));
    // ragged visits with two length classes: the short class's fused launches take their records longest track first (one sort per frame)
    e->visit_order_ok = false;
    // (many filters only: below one workgroup per CU nothing queues, and the sort, the fork and the join are pure launch overhead --
    //  a single sequence went from 1.14 to 1.51 ms per frame with them)
    if (p && np_rec_dev && c->knob.ekf_visit_order != 0 && n_tracks >= 1 && n_tracks <= Ekf::VISIT_SLOTS && (B > (size_t)c->num_cus || c->knob.ekf_visit_order == 2)) {
        // (only where the visits really run as two length classes: 12 stereo poses = 48 rows still ride the short class -- r03 advisor)
        const VisitShape shape = visit_shape(e, np, p->useStereo != 0, p->useLinearTriangulation != 0);
        if (shape.two_class) {
            const int rc = hv::launch_visit_order(c, n_tracks, B, np_rec_dev, 2, shape.np_short, np, e->visit_order, e->visit_long, e->visit_long_count);
            if (rc != HV_OK) return rc;
            e->visit_order_ok = true;
        }
    }
    for (int k = 0; k < n_tracks; ++k) {
        e->visit_slot = k;
        const int rc = visual_track_dev_impl(h, p, np, idx + (size_t)k * B * np, feat + (size_t)k * B * nt * 2, vel + (size_t)k * B * nt * 2,
                                             y + (size_t)k * B * nt * 2, r_gate, r_update, status_dev + (size_t)k * B * 2,
                                             gate_status_dev + (size_t)k * B, chi2_dev ? chi2_dev + (size_t)k * B : nullptr,
                                             pf_dev ? pf_dev + (size_t)k * B * 3 : nullptr, success_counter_dev, max_successful,
                                             np_rec_dev ? np_rec_dev + (size_t)k * B : nullptr);
        e->visit_slot = -1;
        if (rc != HV_OK) { e->visit_order_ok = false; return rc; }
    }
    e->visit_order_ok = false;
    return HV_OK;
}

int hv_ekf_visual_frame_dev(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np, const int *idx, const double *feat, const double *vel,
                            const double *y, double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev,
                            double *pf_dev, int *success_counter_dev, int max_successful)
{
    return visual_frame_dev_impl(h, p, n_tracks, np, nullptr, idx, feat, vel, y, r_gate, r_update, status_dev, gate_status_dev, chi2_dev,
                                 pf_dev, success_counter_dev, max_successful);
}

int hv_ekf_visual_frame_ragged_dev(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np_max, const int *n_poses_dev, const int *idx,
                                   const double *feat, const double *vel, const double *y, double r_gate, double r_update,
                                   int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev, int *success_counter_dev,
                                   int max_successful)
{
    if (!n_poses_dev) return HV_ERR_INVALID;
    return visual_frame_dev_impl(h, p, n_tracks, np_max, n_poses_dev, idx, feat, vel, y, r_gate, r_update, status_dev, gate_status_dev,
                                 chi2_dev, pf_dev, success_counter_dev, max_successful);
}

// Session::trackerVisualUpdate with batchVisualUpdate (or a frame that is not a "full visual update": backend.cpp:1005): see
// ekf_batch_assemble_kernel. <= min(quota, n_tracks) + 1 passes of (prepare + gate of every pending track | assemble | one dense update).
int hv_ekf_visual_frame_batch_dev(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np, const int *np_rec_dev, const int *idx,
                                  const double *feat, const double *vel, const double *y, double r_gate, double r_update,
                                  int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev, int *success_counter_dev,
                                  int max_successful, int max_update_rows)
{
    if (!h || !p || n_tracks < 1 || !success_counter_dev || !status_dev || !gate_status_dev || !idx || !feat || !vel || !y) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    const int ncam = p->useStereo ? 2 : 1, rows = 2 * np * ncam;
    if (max_update_rows <= 0) max_update_rows = e->n;                        // batchVisualUpdateMaxSizeMultiplier 1 (parameter_definitions.c:17)
    if (max_successful <= 0 || max_successful > n_tracks) max_successful = n_tracks;
    // a batch holds at least one whole track (the reference would write past its batch matrix otherwise) and at most stateDim rows (the
    // dense update kernels' limit); the per-filter multiplier of the outlier thresholds would need the tracks of a pass in sequence
    if (max_update_rows < rows || max_update_rows > e->max_rows || n_tracks > 64) return HV_ERR_INVALID;
    if (p->trackOutlierThresholdGrowthFactor != 1.0) return HV_ERR_UNSUPPORTED;
    const size_t B = (size_t)e->batch, rec = B * (size_t)n_tracks;
    if (rec > 8192) return HV_ERR_UNSUPPORTED;                               // (every record's compact Jacobian is resident during a pass)
    const VisitShape shape = visit_shape(e, np, p->useStereo != 0, p->useLinearTriangulation != 0);
    const bool long_build = rows > 48;
    if (long_build ? !(shape.long_ok && c->knob.ekf_long_fused != 0) : !hv::vu_fused_supported(c, e->n, np, p->useStereo != 0, (int)rec)) return HV_ERR_UNSUPPORTED;
    { const int rc_sp = ensure_spec_buffers(e, rec, rows); if (rc_sp != HV_OK) return rc_sp; }
    if (e->b_rows < max_update_rows) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        void *old[] = {e->bH, e->bv, e->brows, e->bany};
        for (void *q : old) if (q) (void)hipFree(q);
        e->bH = e->bv = nullptr; e->brows = nullptr; e->bany = nullptr; e->b_rows = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->bH), sizeof(double) * B * max_update_rows * e->n));
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->bv), sizeof(double) * B * max_update_rows));
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->brows), sizeof(int) * B));
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->bany), B));
        e->b_rows = max_update_rows;
    }
    HV_HIP(c, hipMemsetAsync(success_counter_dev, 0, sizeof(int) * B, c->stream));
    HV_HIP(c, hipMemsetAsync(e->spcursor, 0, sizeof(int) * B, c->stream));
    HV_HIP(c, hipMemsetAsync(e->spepoch, 0xFF, sizeof(int) * rec, c->stream));
    HV_HIP(c, hipMemsetAsync(e->spcursor2, 0xFF, sizeof(int) * B, c->stream));               // carry: -1
    hv::VuPrepareArgs a;
    int rc = vu_fill_args(e, p, np, idx, feat, vel, y, a);
    if (rc != HV_OK) return rc;
    const double ns = e->noise_scale;
    a.H = nullptr; a.v = e->spv; a.f = nullptr; a.pf = pf_dev ? pf_dev : e->sppf; a.status = status_dev; a.active = e->spactive;
    a.gate_status = gate_status_dev; a.success_counter = success_counter_dev; a.max_successful = max_successful;
    a.spec_tracks = n_tracks; a.cursor = e->spcursor; a.epoch = e->spepoch;
    const int *nr_rec = nullptr;
    if (np_rec_dev) { a.np_rec = np_rec_dev; a.rows_out = e->sprows; nr_rec = e->sprows; }
    a.fused = long_build ? 3 : 1; a.Hc = e->spH; a.acol = e->spacol; a.na_max = 7 * np + 1; a.P = e->P;
    a.rd_gate = r_gate * r_gate * ns; a.noise_scale = ns; a.chi2 = chi2_dev;
    hv::BatchAssembleArgs g{};
    g.n = e->n; g.n_tracks = n_tracks; g.batch = e->batch; g.max_rows = max_update_rows; g.max_successful = max_successful;
    g.rows_stride = rows; g.na_max = a.na_max; g.ncam = ncam; g.Hc = e->spH; g.v = e->spv; g.acol = e->spacol; g.nr_rec = nr_rec;
    g.active = e->spactive; g.gate = gate_status_dev; g.cursor = e->spcursor; g.success_counter = success_counter_dev; g.carry = e->spcursor2;
    g.Hd = e->bH; g.vd = e->bv; g.rows_out = e->brows; g.any_out = e->bany;
    const int n_pass = max_successful + 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        rc = hv::launch_vu_prepare(c, a);
        if (rc != HV_OK) return rc;
        hipLaunchKernelGGL(hv::ekf_batch_assemble_kernel, dim3(e->batch), dim3(hv::BATCH_THREADS), 0, c->stream, g);
        HV_HIP(c, hipGetLastError());
        rc = hv::ekf_launch_update(e, max_update_rows, e->n, e->bH, e->bv, nullptr, r_update * r_update * ns, 1, 0, 1, nullptr, nullptr, e->bany,
                                   nullptr, nullptr, 0.0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, e->brows);
        if (rc != HV_OK) return rc;
    }
    return HV_OK;
}

int hv_ekf_visual_track(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel,
                        const double *y, double r_gate, double r_update, int *status, int *gate_status, double *chi2, double *pf)
{
    if (!h || !p || !idx || !feat || !vel || !y || !status || !gate_status || np < 2) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    const size_t B = (size_t)e->batch, nt = (size_t)np * (p->useStereo ? 2 : 1);
    // one staging block, 16-byte aligned sections
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_idx = 0, o_feat = up(o_idx + B * np * sizeof(int)), o_vel = up(o_feat + B * nt * 2 * sizeof(double));
    const size_t o_y = up(o_vel + B * nt * 2 * sizeof(double)), o_st = up(o_y + B * nt * 2 * sizeof(double));
    const size_t o_gs = up(o_st + B * 2 * sizeof(int)), o_chi = up(o_gs + B * sizeof(int)), o_pf = up(o_chi + B * sizeof(double));
    const size_t total = up(o_pf + B * 3 * sizeof(double));
    if (e->vustage_bytes < total) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        if (e->vustage) (void)hipFree(e->vustage);
        e->vustage = nullptr; e->vustage_bytes = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vustage), total));
        e->vustage_bytes = total;
    }
    unsigned char *d = e->vustage;
    HV_HIP(c, hipMemcpyAsync(d + o_idx, idx, B * np * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_feat, feat, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_vel, vel, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_y, y, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    int rc = hv_ekf_visual_track_dev(h, p, np, reinterpret_cast<const int *>(d + o_idx), reinterpret_cast<const double *>(d + o_feat),
                                     reinterpret_cast<const double *>(d + o_vel), reinterpret_cast<const double *>(d + o_y), r_gate,
                                     r_update, reinterpret_cast<int *>(d + o_st), reinterpret_cast<int *>(d + o_gs),
                                     reinterpret_cast<double *>(d + o_chi), reinterpret_cast<double *>(d + o_pf));
    if (rc != HV_OK) return rc;
    HV_HIP(c, hipMemcpyAsync(status, d + o_st, B * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipMemcpyAsync(gate_status, d + o_gs, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (chi2) HV_HIP(c, hipMemcpyAsync(chi2, d + o_chi, B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pf) HV_HIP(c, hipMemcpyAsync(pf, d + o_pf, B * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

// batch_rows: 0 = the sequential visit loop (visual_frame_dev_impl); != 0 = the batchVisualUpdate loop with this max_update_rows (< 0: the
// library's default, stateDim)
static int visual_frame_host_impl(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np, const int *n_poses, const int *idx,
                                  const double *feat, const double *vel,
                        const double *y, double r_gate, double r_update, int *status, int *gate_status, double *chi2, double *pf,
                        int *success_count, int max_successful, int batch_rows = 0)
{
    if (!h || !p || !idx || !feat || !vel || !y || !status || !gate_status || np < 2 || n_tracks < 1) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    const size_t B = (size_t)e->batch * n_tracks, nt = (size_t)np * (p->useStereo ? 2 : 1);
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_idx = 0, o_feat = up(o_idx + B * np * sizeof(int)), o_vel = up(o_feat + B * nt * 2 * sizeof(double));
    const size_t o_y = up(o_vel + B * nt * 2 * sizeof(double)), o_st = up(o_y + B * nt * 2 * sizeof(double));
    const size_t o_gs = up(o_st + B * 2 * sizeof(int)), o_chi = up(o_gs + B * sizeof(int)), o_pf = up(o_chi + B * sizeof(double));
    const size_t o_cnt = up(o_pf + B * 3 * sizeof(double)), o_np = up(o_cnt + (size_t)e->batch * sizeof(int));
    const size_t total = up(o_np + B * sizeof(int));
    if (e->vustage_bytes < total) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        if (e->vustage) (void)hipFree(e->vustage);
        e->vustage = nullptr; e->vustage_bytes = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vustage), total));
        e->vustage_bytes = total;
    }
    unsigned char *d = e->vustage;
    HV_HIP(c, hipMemcpyAsync(d + o_idx, idx, B * np * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_feat, feat, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_vel, vel, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_y, y, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (n_poses) HV_HIP(c, hipMemcpyAsync(d + o_np, n_poses, B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int *d_np = n_poses ? reinterpret_cast<const int *>(d + o_np) : nullptr;
    const int rc = batch_rows == 0
        ? visual_frame_dev_impl(h, p, n_tracks, np, d_np,
                                           reinterpret_cast<const int *>(d + o_idx), reinterpret_cast<const double *>(d + o_feat),
                                           reinterpret_cast<const double *>(d + o_vel), reinterpret_cast<const double *>(d + o_y), r_gate, r_update,
                                           reinterpret_cast<int *>(d + o_st), reinterpret_cast<int *>(d + o_gs), reinterpret_cast<double *>(d + o_chi),
                                           reinterpret_cast<double *>(d + o_pf), reinterpret_cast<int *>(d + o_cnt), max_successful)
        : hv_ekf_visual_frame_batch_dev(h, p, n_tracks, np, d_np, reinterpret_cast<const int *>(d + o_idx), reinterpret_cast<const double *>(d + o_feat),
                                        reinterpret_cast<const double *>(d + o_vel), reinterpret_cast<const double *>(d + o_y), r_gate, r_update,
                                        reinterpret_cast<int *>(d + o_st), reinterpret_cast<int *>(d + o_gs), reinterpret_cast<double *>(d + o_chi),
                                        reinterpret_cast<double *>(d + o_pf), reinterpret_cast<int *>(d + o_cnt), max_successful,
                                        batch_rows < 0 ? 0 : batch_rows);
    if (rc != HV_OK) return rc;
    HV_HIP(c, hipMemcpyAsync(status, d + o_st, B * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipMemcpyAsync(gate_status, d + o_gs, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (chi2) HV_HIP(c, hipMemcpyAsync(chi2, d + o_chi, B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pf) HV_HIP(c, hipMemcpyAsync(pf, d + o_pf, B * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (success_count) HV_HIP(c, hipMemcpyAsync(success_count, d + o_cnt, (size_t)e->batch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    int flags = 0;
    const int rc2 = hv_ekf_frame_error(h, &flags);                // the hand-shake form of the speculative pass never fails silently
    if (rc2 != HV_OK) return rc2;
    return flags ? HV_ERR_TIMEOUT : HV_OK;
}

int hv_ekf_visual_frame(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np, const int *idx, const double *feat, const double *vel,
                        const double *y, double r_gate, double r_update, int *status, int *gate_status, double *chi2, double *pf,
                        int *success_count, int max_successful)
{
    return visual_frame_host_impl(h, p, n_tracks, np, nullptr, idx, feat, vel, y, r_gate, r_update, status, gate_status, chi2, pf,
                                  success_count, max_successful);
}

int hv_ekf_visual_frame_ragged(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np_max, const int *n_poses, const int *idx,
                               const double *feat, const double *vel, const double *y, double r_gate, double r_update, int *status,
                               int *gate_status, double *chi2, double *pf, int *success_count, int max_successful)
{
    if (!n_poses) return HV_ERR_INVALID;
    return visual_frame_host_impl(h, p, n_tracks, np_max, n_poses, idx, feat, vel, y, r_gate, r_update, status, gate_status, chi2, pf,
                                  success_count, max_successful);
}

int hv_ekf_visual_frame_batch(hv_ekf *h, const hv_vu_params *p, int n_tracks, int np_max, const int *n_poses, const int *idx,
                              const double *feat, const double *vel, const double *y, double r_gate, double r_update, int *status,
                              int *gate_status, double *chi2, double *pf, int *success_count, int max_successful, int max_update_rows)
{
    return visual_frame_host_impl(h, p, n_tracks, np_max, n_poses, idx, feat, vel, y, r_gate, r_update, status, gate_status, chi2, pf,
                                  success_count, max_successful, max_update_rows > 0 ? max_update_rows : -1);
}

// host-pointer form of hv_ekf_visual_track_hybrid_dev: one track per filter, arrays [batch]...
int hv_ekf_visual_track_hybrid(hv_ekf *h, const hv_vu_params *p, int np, const int *idx, const double *feat, const double *vel, const double *y,
                               const int *map_update, const int *map_offer, double r_gate, double r_update, int *status, int *gate_status,
                               double *chi2, double *pf)
{
    if (!h || !p || !idx || !feat || !vel || !y || !status || !gate_status || np < 2) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    const size_t B = (size_t)e->batch, nt = (size_t)np * (p->useStereo ? 2 : 1);
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_idx = 0, o_feat = up(o_idx + B * np * sizeof(int)), o_vel = up(o_feat + B * nt * 2 * sizeof(double));
    const size_t o_y = up(o_vel + B * nt * 2 * sizeof(double)), o_st = up(o_y + B * nt * 2 * sizeof(double));
    const size_t o_gs = up(o_st + B * 2 * sizeof(int)), o_chi = up(o_gs + B * sizeof(int)), o_pf = up(o_chi + B * sizeof(double));
    const size_t o_mu = up(o_pf + B * 3 * sizeof(double)), o_mo = up(o_mu + B * sizeof(int)), total = up(o_mo + B * sizeof(int));
    if (e->vustage_bytes < total) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        if (e->vustage) (void)hipFree(e->vustage);
        e->vustage = nullptr; e->vustage_bytes = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&e->vustage), total));
        e->vustage_bytes = total;
    }
    unsigned char *d = e->vustage;
    HV_HIP(c, hipMemcpyAsync(d + o_idx, idx, B * np * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_feat, feat, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_vel, vel, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_y, y, B * nt * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (map_update) HV_HIP(c, hipMemcpyAsync(d + o_mu, map_update, B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (map_offer) HV_HIP(c, hipMemcpyAsync(d + o_mo, map_offer, B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int rc = hv_ekf_visual_track_hybrid_dev(h, p, np, reinterpret_cast<const int *>(d + o_idx), reinterpret_cast<const double *>(d + o_feat),
                                                  reinterpret_cast<const double *>(d + o_vel), reinterpret_cast<const double *>(d + o_y),
                                                  map_update ? reinterpret_cast<const int *>(d + o_mu) : nullptr,
                                                  map_offer ? reinterpret_cast<const int *>(d + o_mo) : nullptr, r_gate, r_update,
                                                  reinterpret_cast<int *>(d + o_st), reinterpret_cast<int *>(d + o_gs),
                                                  reinterpret_cast<double *>(d + o_chi), reinterpret_cast<double *>(d + o_pf));
    if (rc != HV_OK) return rc;
    HV_HIP(c, hipMemcpyAsync(status, d + o_st, B * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipMemcpyAsync(gate_status, d + o_gs, B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (chi2) HV_HIP(c, hipMemcpyAsync(chi2, d + o_chi, B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pf) HV_HIP(c, hipMemcpyAsync(pf, d + o_pf, B * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

// insertMapPoint (ekf.cpp:911-921) on the resident state of one filter: nothing but the three coordinates crosses the bus
namespace hv { namespace {
__global__ __launch_bounds__(256) void ekf_insert_map_point_kernel(int n, int off, double *m, double *P, double x, double y, double z)
{
    const int t = threadIdx.x;
    for (int i = t; i < 3 * n; i += 256) {
        const int k = i / n, j = i - k * n;
        P[(size_t)(off + k) * n + j] = 0.0;
        P[(size_t)j * n + off + k] = 0.0;
    }
    __syncthreads();
    if (t < 3) { P[(size_t)(off + t) * n + off + t] = 1e6; m[off + t] = t == 0 ? x : t == 1 ? y : z; }
}
} }

int hv_ekf_insert_map_point(hv_ekf *h, int b, int map_index, const double *pf)
{
    if (!h || !pf || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (map_index < 0 || 3 * (map_index + 1) > e->map_dim) return HV_ERR_INVALID;
    const int off = e->n - e->map_dim + 3 * map_index;
    hipLaunchKernelGGL(hv::ekf_insert_map_point_kernel, dim3(1), dim3(256), 0, c->stream, e->n, off, e->m + (size_t)b * e->n,
                       e->P + (size_t)b * e->n * e->n, pf[0], pf[1], pf[2]);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

int hv_ekf_get_map_point(hv_ekf *h, int b, int map_index, double *pf)
{
    if (!h || !pf || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (map_index < 0 || 3 * (map_index + 1) > e->map_dim) return HV_ERR_INVALID;
    HV_HIP(c, hipMemcpyAsync(pf, e->m + (size_t)b * e->n + e->n - e->map_dim + 3 * map_index, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_frame_error(hv_ekf *h, int *flags)
{
    if (!h || !flags) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    HV_HIP(c, hipMemcpyAsync(flags, e->err_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    if (*flags) HV_HIP(c, hipMemsetAsync(e->err_dev, 0, sizeof(int), c->stream));
    return HV_OK;
}

/* developer aid (not in the public header): phase time stamps of the last update kernel */
int hv_debug_ekf_phase_stamps(hv_ekf *h, long long *out8)
{
    if (!h || !out8) return HV_ERR_INVALID;
    Ctx *c = h->e.c;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    HV_HIP(c, hipMemcpyFromSymbol(out8, HIP_SYMBOL(hv::g_phase_stamp), sizeof(long long) * 16));
    return HV_OK;
}

int hv_ekf_state_dim(const hv_ekf *h) { return h ? h->e.n : HV_ERR_INVALID; }
int hv_ekf_batch(const hv_ekf *h) { return h ? h->e.batch : HV_ERR_INVALID; }

int hv_ekf_set_state(hv_ekf *h, int b, const double *m, const double *P)
{
    if (!h || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c; const size_t n = e->n;
    if (m) HV_HIP(c, hipMemcpyAsync(e->m + b * n, m, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (P) HV_HIP(c, hipMemcpyAsync(e->P + b * n * n, P, sizeof(double) * n * n, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_get_state(hv_ekf *h, int b, double *m, double *P)
{
    if (!h || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c; const size_t n = e->n;
    if (m) HV_HIP(c, hipMemcpyAsync(m, e->m + b * n, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    if (P) HV_HIP(c, hipMemcpyAsync(P, e->P + b * n * n, sizeof(double) * n * n, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_get_means(hv_ekf *h, double *m_all)
{
    if (!h || !m_all) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    HV_HIP(c, hipMemcpyAsync(m_all, e->m, sizeof(double) * e->n * e->batch, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_set_process_noise(hv_ekf *h, int b, const double *Q)
{
    if (!h || !Q || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ctx *c = h->e.c;
    HV_HIP(c, hipMemcpyAsync(h->e.Q + (size_t)b * 144, Q, sizeof(double) * 144, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_get_process_noise(hv_ekf *h, int b, double *Q)
{
    if (!h || !Q || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ctx *c = h->e.c;
    HV_HIP(c, hipMemcpyAsync(Q, h->e.Q + (size_t)b * 144, sizeof(double) * 144, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_get_dydx(hv_ekf *h, int b, double *F)
{
    if (!h || !F || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ctx *c = h->e.c;
    HV_HIP(c, hipMemcpyAsync(F, h->e.dydx + (size_t)b * 400, sizeof(double) * 400, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_device_pointers(hv_ekf *h, double **m_dev, double **P_dev)
{
    if (!h) return HV_ERR_INVALID;
    if (m_dev) *m_dev = h->e.m;
    if (P_dev) *P_dev = h->e.P;
    return HV_OK;
}

static int predict_common(Ekf *e, const double *dt_dev, const double *gyro_dev, const double *acc_dev,
                          double dt0, const double *g0, const double *a0, int nsteps = 1)
{
    Ctx *c = e->c;
    hv::PredictArgs a{};
    a.n = e->n; a.batch = e->batch; a.m = e->m; a.P = e->P; a.Q = e->Q; a.dydx = e->dydx;
    a.dt = dt_dev; a.gyro = gyro_dev; a.acc = acc_dev; a.dt0 = dt0; a.nsteps = nsteps;
    for (int i = 0; i < 3; i++) { a.g0[i] = g0 ? g0[i] : 0.0; a.a0[i] = a0 ? a0[i] : 0.0; }
    a.noise_scale = e->noise_scale; a.gravity = e->par.gravity;
    a.baa = e->par.noiseProcessBAA; a.baa_rev = e->par.noiseProcessBAARev;
    a.bga = e->par.noiseProcessBGA; a.bga_rev = e->par.noiseProcessBGARev;
    hv::ScopedKernelTime tm(c, HV_K_EKF_PREDICT);
    // knob ekf_predict_chain: 1 (late r06) = launches of three or more samples run ekf_predict_chain_kernel (the mean recursion of the samples
    // on one wavefront, F / L of a chunk of samples in one pass); 2 = every launch; 0 = never (ekf_predict_kernel: nine barrier-separated
    // stages per sample). Bit-identical results. Measured (profiles/r06/predict_chain_ab.txt): 10 samples 53.2 -> 50.3 us for one filter,
    // 139 -> 129 us at 1024 filters; ONE sample 13.4 -> 15.0 us, hence the rule
    const int chain = c->knob.ekf_predict_chain;
    if (chain == 2 || (chain == 1 && nsteps >= 3)) hipLaunchKernelGGL(hv::ekf_predict_chain_kernel, dim3(e->batch), dim3(256), 0, c->stream, a);
    else                                hipLaunchKernelGGL(hv::ekf_predict_kernel, dim3(e->batch), dim3(256), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

int hv_ekf_predict(hv_ekf *h, const double *dt, const double *gyro, const double *acc)
{
    if (!h || !dt || !gyro || !acc) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (e->batch == 1) return predict_common(e, nullptr, nullptr, nullptr, dt[0], gyro, acc);   // immediates, no copy
    const size_t B = e->batch;
    HV_HIP(c, hipMemcpyAsync(e->simu, dt, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(e->simu + B, gyro, sizeof(double) * 3 * B, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(e->simu + 4 * B, acc, sizeof(double) * 3 * B, hipMemcpyHostToDevice, c->stream));
    return predict_common(e, e->simu, e->simu + B, e->simu + 4 * B, 0.0, nullptr, nullptr);
}

int hv_ekf_predict_dev(hv_ekf *h, const double *dt_dev, const double *gyro_dev, const double *acc_dev)
{
    if (!h || !dt_dev || !gyro_dev || !acc_dev) return HV_ERR_INVALID;
    return predict_common(&h->e, dt_dev, gyro_dev, acc_dev, 0.0, nullptr, nullptr);
}

int hv_ekf_predict_n(hv_ekf *h, int n_samples, const double *dt, const double *gyro, const double *acc)
{
    if (!h || n_samples < 1 || n_samples > HV_EKF_MAX_PREDICT_SAMPLES || !dt || !gyro || !acc) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (n_samples == 1 && e->batch == 1) return predict_common(e, nullptr, nullptr, nullptr, dt[0], gyro, acc);
    const size_t nB = (size_t)n_samples * e->batch;
    double *d_dt = e->simu, *d_g = e->simu + nB, *d_a = e->simu + 4 * nB;
    HV_HIP(c, hipMemcpyAsync(d_dt, dt, sizeof(double) * nB, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d_g, gyro, sizeof(double) * 3 * nB, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d_a, acc, sizeof(double) * 3 * nB, hipMemcpyHostToDevice, c->stream));
    return predict_common(e, d_dt, d_g, d_a, 0.0, nullptr, nullptr, n_samples);
}

int hv_ekf_predict_n_dev(hv_ekf *h, int n_samples, const double *dt_dev, const double *gyro_dev, const double *acc_dev)
{
    if (!h || n_samples < 1 || !dt_dev || !gyro_dev || !acc_dev) return HV_ERR_INVALID;
    return predict_common(&h->e, dt_dev, gyro_dev, acc_dev, 0.0, nullptr, nullptr, n_samples);
}

static int stage_update_inputs(Ekf *e, int nr, int l, const double *H, const double *v, const double *rdiag,
                               const unsigned char *active)
{
    Ctx *c = e->c; const size_t B = e->batch;
    if ((size_t)nr * l * B > e->sH_cap) return HV_ERR_INVALID;
    HV_HIP(c, hipMemcpyAsync(e->sH, H, sizeof(double) * nr * l * B, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(e->sv, v, sizeof(double) * nr * B, hipMemcpyHostToDevice, c->stream));
    if (rdiag) HV_HIP(c, hipMemcpyAsync(e->sr, rdiag, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    if (active) HV_HIP(c, hipMemcpyAsync(e->sactive, active, B, hipMemcpyHostToDevice, c->stream));
    return HV_OK;
}

int hv_ekf_update(hv_ekf *h, int nr, int l, const double *H, const double *y, const double *r_diag,
                  const unsigned char *active, int normalize_all)
{
    if (!h || !H || !y || !r_diag) return HV_ERR_INVALID;
    Ekf *e = &h->e;
    if (nr < 1 || nr > e->max_rows || l < 1 || l > e->n) return HV_ERR_INVALID;
    int rc = stage_update_inputs(e, nr, l, H, y, r_diag, active);
    if (rc != HV_OK) return rc;
    return hv::ekf_launch_update(e, nr, l, e->sH, e->sv, e->sr, 0.0, 1, 1, normalize_all, nullptr, nullptr,
                                 active ? e->sactive : nullptr);
}

int hv_ekf_visual_gate(hv_ekf *h, int nr, int l, const double *H, const double *v, double r, double *chi2, int *status)
{
    if (!h || !H || !v) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    if (nr < 1 || nr > e->max_rows || l < 1 || l > e->n) return HV_ERR_INVALID;
    int rc = stage_update_inputs(e, nr, l, H, v, nullptr, nullptr);
    if (rc != HV_OK) return rc;
    rc = hv::ekf_launch_update(e, nr, l, e->sH, e->sv, nullptr, r * r * e->noise_scale, 0, 0, 0, e->schi2, e->sstatus, nullptr);
    if (rc != HV_OK) return rc;
    if (chi2) HV_HIP(c, hipMemcpyAsync(chi2, e->schi2, sizeof(double) * e->batch, hipMemcpyDeviceToHost, c->stream));
    if (status) HV_HIP(c, hipMemcpyAsync(status, e->sstatus, sizeof(int) * e->batch, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_ekf_visual_update(hv_ekf *h, int nr, int l, const double *H, const double *v, double r, const unsigned char *active)
{
    if (!h || !H || !v) return HV_ERR_INVALID;
    Ekf *e = &h->e;
    if (nr < 1 || nr > e->max_rows || l < 1 || l > e->n) return HV_ERR_INVALID;
    int rc = stage_update_inputs(e, nr, l, H, v, nullptr, active);
    if (rc != HV_OK) return rc;
    return hv::ekf_launch_update(e, nr, l, e->sH, e->sv, nullptr, r * r * e->noise_scale, 1, 0, 1, nullptr, nullptr,
                                 active ? e->sactive : nullptr);
}

int hv_ekf_visual_dev(hv_ekf *h, int nr, int l, const double *H_dev, const double *v_dev, double r, int mode,
                      double *chi2_dev, int *status_dev)
{
    if (!h || !H_dev || !v_dev || mode < 0 || mode > 2) return HV_ERR_INVALID;
    Ekf *e = &h->e;
    const int stream_gate = e->c->knob.ekf_stream_gate;
    if (mode == 0 && (stream_gate == 1 || (stream_gate < 0 && e->batch > 256))) {      // gate only, many filters: two per CU
        bool done = false;
        const int rc = hv::ekf_launch_gate_stream(e, nr, l, H_dev, v_dev, r * r * e->noise_scale, chi2_dev, status_dev, nullptr, nullptr, 0, &done);
        if (rc != HV_OK || done) return rc;
    }
    return hv::ekf_launch_update(e, nr, l, H_dev, v_dev, nullptr, r * r * e->noise_scale, mode, 0, 1, chi2_dev, status_dev, nullptr);
}

int hv_ekf_augment(hv_ekf *h, const int *discarded, const unsigned char *active)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hv::AugmentArgs a{};
    a.n = e->n; a.cam_poses = e->cam; a.map_dim = e->map_dim;
    a.m = e->m; a.P = e->P; a.P1 = e->P1; a.m1 = e->m1;
    a.dropped0 = -1;
    if (discarded) {
        if (e->batch == 1) a.dropped0 = discarded[0];
        else { HV_HIP(c, hipMemcpyAsync(e->sdrop, discarded, sizeof(int) * e->batch, hipMemcpyHostToDevice, c->stream)); a.dropped = e->sdrop; }
    }
    if (active) { HV_HIP(c, hipMemcpyAsync(e->sactive, active, e->batch, hipMemcpyHostToDevice, c->stream)); a.active = e->sactive; }
    a.q_pos = e->par.noiseInitialPosTrail * e->par.noiseInitialPosTrail * e->noise_scale;
    a.q_ori = e->par.noiseInitialOriTrail * e->par.noiseInitialOriTrail * e->noise_scale;
    a.rd = e->par.augmentR * e->noise_scale;
    const size_t shmem = sizeof(double) * (3 * hv::POSE * e->n + 2 * hv::POSE * hv::POSE + hv::POSE + 1 + (hv::AUG_THREADS / 64) * 16 * 17);
    if (shmem > 64 * 1024) {                 // state vectors with map points (n > ~190): beyond the default dynamic-LDS limit
        if (shmem > 158 * 1024) return HV_ERR_UNSUPPORTED;
        static bool aug_attr_set_dev[64] = {};
        bool &aug_attr_set = aug_attr_set_dev[c->p.device & 63];
        if (!aug_attr_set) {
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(hv::ekf_augment_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
            aug_attr_set = true;
        }
    }
    hv::ScopedKernelTime tm(c, HV_K_EKF_AUGMENT);
    hipLaunchKernelGGL(hv::ekf_augment_kernel<false>, dim3(e->batch), dim3(hv::AUG_THREADS), shmem, c->stream, a);
    HV_HIP(c, hipGetLastError());
    std::swap(e->P, e->P1);                 // the kernel wrote the new covariance to the other buffer
    return HV_OK;
}

static int augment_dev_impl(hv_ekf *h, const int *discarded_dev, const unsigned char *active_dev, bool sym_input)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hv::AugmentArgs a{};
    a.n = e->n; a.cam_poses = e->cam; a.map_dim = e->map_dim;
    a.m = e->m; a.P = e->P; a.P1 = e->P1; a.m1 = e->m1;
    a.dropped0 = -1; a.dropped = discarded_dev; a.active = active_dev;
    a.q_pos = e->par.noiseInitialPosTrail * e->par.noiseInitialPosTrail * e->noise_scale;
    a.q_ori = e->par.noiseInitialOriTrail * e->par.noiseInitialOriTrail * e->noise_scale;
    a.rd = e->par.augmentR * e->noise_scale;
    const size_t shmem = sizeof(double) * (3 * hv::POSE * e->n + 2 * hv::POSE * hv::POSE + hv::POSE + 1 + (hv::AUG_THREADS / 64) * 16 * 17);
    if (shmem > 64 * 1024) return HV_ERR_UNSUPPORTED;        // map-point states: use hv_ekf_augment (it raises the LDS limit)
    hv::ScopedKernelTime tm(c, HV_K_EKF_AUGMENT);
    if (sym_input) hipLaunchKernelGGL(hv::ekf_augment_kernel<true>, dim3(e->batch), dim3(hv::AUG_THREADS), shmem, c->stream, a);
    else           hipLaunchKernelGGL(hv::ekf_augment_kernel<false>, dim3(e->batch), dim3(hv::AUG_THREADS), shmem, c->stream, a);
    HV_HIP(c, hipGetLastError());
    std::swap(e->P, e->P1);
    return HV_OK;
}

// host-pointer form of hv_ekf_symmetrize_augment_dev (ABI 4)
int hv_ekf_symmetrize_augment(hv_ekf *h, const int *discarded, const unsigned char *active)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    const int *d_drop = nullptr; const unsigned char *d_act = nullptr;
    if (discarded) { HV_HIP(c, hipMemcpyAsync(e->sdrop, discarded, sizeof(int) * e->batch, hipMemcpyHostToDevice, c->stream)); d_drop = e->sdrop; }
    if (active) { HV_HIP(c, hipMemcpyAsync(e->sactive, active, e->batch, hipMemcpyHostToDevice, c->stream)); d_act = e->sactive; }
    const int rc = augment_dev_impl(h, d_drop, d_act, true);
    if (rc != HV_ERR_UNSUPPORTED) return rc;
    // (map-point states: the folded kernel's LDS limit; the two calls it stands for)
    const int rc2 = hv_ekf_symmetrize(h);
    return rc2 != HV_OK ? rc2 : hv_ekf_augment(h, discarded, active);
}

int hv_ekf_augment_dev(hv_ekf *h, const int *discarded_dev, const unsigned char *active_dev) { return augment_dev_impl(h, discarded_dev, active_dev, false); }
int hv_ekf_symmetrize_augment_dev(hv_ekf *h, const int *discarded_dev, const unsigned char *active_dev) { return augment_dev_impl(h, discarded_dev, active_dev, true); }

int hv_ekf_undo_augment(hv_ekf *h, const unsigned char *active)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hv::ShiftArgs a{e->n, e->map_dim, e->m, e->P, e->P1, e->m1, nullptr};
    if (active) { HV_HIP(c, hipMemcpyAsync(e->sactive, active, e->batch, hipMemcpyHostToDevice, c->stream)); a.active = e->sactive; }
    hv::ScopedKernelTime tm(c, HV_K_EKF_AUGMENT);
    hipLaunchKernelGGL(hv::ekf_unaugment_kernel, dim3(e->batch), dim3(1024), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    std::swap(e->P, e->P1);
    return HV_OK;
}

int hv_ekf_symmetrize(hv_ekf *h)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hipLaunchKernelGGL(hv::ekf_symmetrize_kernel, dim3(e->batch), dim3(1024), 0, c->stream, e->n, e->P);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

int hv_ekf_normalize_quaternions(hv_ekf *h, int only_current)
{
    if (!h) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hipLaunchKernelGGL(hv::ekf_normalize_kernel, dim3(e->batch), dim3(64), 0, c->stream, e->n, e->map_dim, e->m, only_current);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

int hv_ekf_transform(hv_ekf *h, int b, const double *pC3x3, const double *qC4x4, const double *translation3)
{
    if (!h || !pC3x3 || !qC4x4 || !translation3 || b < 0 || b >= h->e.batch) return HV_ERR_INVALID;
    Ekf *e = &h->e; Ctx *c = e->c;
    hv::TransformArgs a{};
    a.n = e->n; a.cam_poses = e->cam;
    a.m = e->m + (size_t)b * e->n; a.P = e->P + (size_t)b * e->n * e->n;
    for (int i = 0; i < 9; i++) a.pC[i] = pC3x3[i];
    for (int i = 0; i < 16; i++) a.qC[i] = qC4x4[i];
    for (int i = 0; i < 3; i++) a.tr[i] = translation3[i];
    hipLaunchKernelGGL(hv::ekf_transform_kernel, dim3(8), dim3(256), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // extern "C"
