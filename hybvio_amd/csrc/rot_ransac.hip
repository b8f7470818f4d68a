// 2-point rotation RANSAC on the LK output.
//
// Reference: src/tracker/rot_ransac.cpp:41-120 (RotRansac::fit), :122-126 (withinInlierThreshold), :132-160
// (solveRotation), src/tracker/ransac_pipeline.cpp:91-93,197-216 (threshold, doRansac2), src/tracker/camera.cpp:93-221
// (pinhole model), :264-398 (fisheye model).
//
// One workgroup per point set (one frame of one sequence). The 100 hypotheses of the reference loop are independent
// once their index pairs are known, so the host draws the pairs (rng() % n, to keep std::mt19937 in step) and the
// kernel evaluates all hypotheses at once: thread k solves rotation k, then every (point, hypothesis) pair is tested,
// counts are integer LDS atomics, and the loop's "first maximum / stop at the first all-inlier hypothesis" is taken from
// the counts afterwards. The refit accumulates H over the inliers in index order in binary32 (one lane per entry of H),
// so that the result does not depend on the thread count. Arithmetic follows oracle/rot_ransac_oracle.c operation by
// operation (the library is built without FMA contraction): statuses, counts and R are bit-identical for the pinhole
// model; the fisheye model calls sin / cos / acos, whose last bit may differ from the host's libm.
#include "hv_internal.hpp"

#include <algorithm>
#include <cmath>

namespace hv {
namespace {

constexpr int RT_MANY = 256;           // workgroup size with many frames in flight (8 frames per CU)
constexpr int RT_FEW = 1024;           // a handful of frames: the 20 000 (point, hypothesis) tests of ONE frame spread 4x wider (91 -> ~30 us)
constexpr int MAX_PTS = 1024;
constexpr int HYP = 100;              // ROT_RANSAC_MAX_ITERS (rot_ransac.cpp:6)
// r05, a single sequence's frame (<= num_cus / RT_SPLIT_GROUPS sets in flight): the hypotheses of ONE set spread over RT_SPLIT_GROUPS
// workgroups -- each forms the rays of every point (cheap) and counts the inliers of its own HYP / RT_SPLIT_GROUPS hypotheses (the 20 000
// point x hypothesis tests are what a lone workgroup spends its time on); counts, validity and rotations meet in a small global record
// and the workgroup that arrives last (a ticket) runs the reference loop's bookkeeping, the refit and the final classification.
// Integer counts and the same rotations: the results are the lone workgroup's, bit for bit.
constexpr int RT_SPLIT_GROUPS = 25, RT_SPLIT = 256;
static_assert(HYP % RT_SPLIT_GROUPS == 0, "whole groups of hypotheses");
constexpr size_t SPLIT_REC_BYTES = sizeof(int) * (2 * HYP + 4) + sizeof(float) * 9 * HYP;   // per set: counts, valid, ticket (+pad), rotations

struct RansacArgs {
    int max_points;
    const int *n_points;              // [sets]
    const float *c1, *c2;             // [sets][max_points][2]
    const int *pairs;                 // [sets][HYP][2], or NULL when draws is given
    const uint32_t *draws;            // [sets][2 HYP] raw generator outputs: pair k = (draws[2k] % n, draws[2k+1] % n)
    const uint8_t *lk_status;         // optional [sets][max_points]: only features with lk_status == lk_tracked take part
    int lk_tracked;                   // (c1 / c2 / status are then indexed by the ORIGINAL feature number)
    float threshold_pow2;
    int *status;                      // [sets][max_points]: 0 TRACKED / 3 RANSAC_OUTLIER
    float *R;                         // [sets][9]
    int *summary;                     // [sets][2]: bestInlierCount, hypotheses visited by the reference loop
    unsigned char *split;             // split launches: [sets] records of SPLIT_REC_BYTES (tickets zero between launches)
    hv_camera_model cam1, cam2;
};

// ---- camera models (camera.cpp), double precision, the oracle's operation order ----
__device__ void pin_distort(const hv_camera_model &c, double &x, double &y, double *J)
{
    if (!c.distortion_enabled) { J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1; return; }
    const double *k = c.coeffs, X = x, Y = y, r2 = X * X + Y * Y;
    const double theta = 1 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]));
    const double dth = k[0] + r2 * (k[1] * 2 + r2 * k[2] * 3);
    J[0] = theta + X * dth * 2 * X; J[1] = X * dth * 2 * Y;
    J[2] = Y * dth * 2 * X;         J[3] = theta + Y * dth * 2 * Y;
    x = X * theta; y = Y * theta;
}

__device__ double fish_distort(const hv_camera_model &c, double theta, double *der)
{
    if (!c.distortion_enabled) { if (der) *der = 1.0; return theta; }
    const double *k = c.coeffs, t = theta, t2 = t * t;
    if (der) *der = 1 + 3 * t2 * (k[0] + 5.0 / 3 * t2 * (k[1] + 7.0 / 5 * t2 * (k[2] + 9.0 / 7 * t2 * k[3])));
    return t * (1 + t2 * (k[0] + t2 * (k[1] + t2 * (k[2] + t2 * k[3]))));
}

__device__ double fish_newton(const hv_camera_model &c, double r, double theta0)
{
    const double eps = 0.01 / ((c.fx + c.fy) * 0.5);
    double theta = theta0, d;
    for (int it = 0; it < 20; ++it) {
        const double dr = fish_distort(c, theta, &d) - r, dt = dr / d;
        theta -= dt;
        if (fabs(dt) < eps) return theta > 0.0 ? theta : 0.0;
    }
    return -1;
}

__device__ void pixel_to_ray(const hv_camera_model &c, double px, double py, double *ray)
{
    if (c.kind == 0) {                                                       // camera.cpp:169-180
        double x = (px - c.ppx) / c.fx, y = (py - c.ppy) / c.fy;
        if (c.distortion_enabled) {                                          // :108-123 Newton
            const double dx = x, dy = y;
            double nrm;
            int it = 0;
            do {
                double qx = x, qy = y, J[4];
                pin_distort(c, qx, qy, J);
                const double id = 1.0 / (J[0] * J[3] - J[1] * J[2]);
                const double ex = dx - qx, ey = dy - qy;
                const double sx = (J[3] * id) * ex + (-J[1] * id) * ey, sy = (-J[2] * id) * ex + (J[0] * id) * ey;
                x += sx; y += sy;
                nrm = sqrt(sx * sx + sy * sy);
            } while (nrm > 1e-5 && ++it < 100);
        }
        const double n = sqrt(x * x + y * y + 1.0);
        double r[3] = {x / n, y / n, 1.0 / n};
        if (c.rotation_enabled) {
            const double *R = c.rotation;
            const double t0 = R[0] * r[0] + R[1] * r[1] + R[2] * r[2], t1 = R[3] * r[0] + R[4] * r[1] + R[5] * r[2],
                         t2 = R[6] * r[0] + R[7] * r[1] + R[8] * r[2];
            r[0] = t0; r[1] = t1; r[2] = t2;
        }
        ray[0] = r[0]; ray[1] = r[1]; ray[2] = r[2];
        return;
    }
    const double *Ki = c.kinv;                                               // :353-375
    const double u = Ki[0] * px + Ki[1] * py + Ki[2], v = Ki[3] * px + Ki[4] * py + Ki[5];
    const double r = sqrt(u * u + v * v), dxn = u / r, dyn = v / r;
    double theta = r;
    if (r > c.max_r) theta = c.max_theta;
    else if (c.distortion_enabled) {
        const int n = c.n_table;
        double f = r / c.max_r; if (!(f > 0.0)) f = 0.0;
        int i = (int)(f * (double)n); if (i > n - 1) i = n - 1;
        const double th = fish_newton(c, r, c.table[i]);
        theta = th < 0 ? r : th;
    }
    const double s = sin(theta);
    ray[0] = s * dxn; ray[1] = s * dyn; ray[2] = cos(theta);
}

__device__ bool ray_to_pixel(const hv_camera_model &c, const double *ray0, double *pix)
{
    if (c.kind == 0) {                                                       // camera.cpp:182-205
        double r[3] = {ray0[0], ray0[1], ray0[2]};
        if (c.rotation_enabled) {
            const double *R = c.rotation;
            const double t0 = R[0] * r[0] + R[3] * r[1] + R[6] * r[2], t1 = R[1] * r[0] + R[4] * r[1] + R[7] * r[2],
                         t2 = R[2] * r[0] + R[5] * r[1] + R[8] * r[2];
            r[0] = t0; r[1] = t1; r[2] = t2;
        }
        if (r[2] <= 0) return false;
        const double iz = 1.0 / r[2];
        double x = r[0] * iz, y = r[1] * iz, J[4];
        pin_distort(c, x, y, J);
        pix[0] = c.fx * x + 0.0 * y + c.ppx * (r[2] * iz);
        pix[1] = 0.0 * x + c.fy * y + c.ppy * (r[2] * iz);
        return true;
    }
    if (ray0[2] <= 0) return false;                                          // :377-398
    const double inv = 1.0 / sqrt(ray0[0] * ray0[0] + ray0[1] * ray0[1] + ray0[2] * ray0[2]);
    const double theta = acos(ray0[2] * inv);
    if (theta > c.max_theta) return false;
    const double r = fish_distort(c, theta, nullptr);
    const double n2 = ray0[0] * ray0[0] + ray0[1] * ray0[1];
    double dx = ray0[0], dy = ray0[1];
    if (n2 > 0) { const double n = sqrt(n2); dx /= n; dy /= n; }
    const double u = r * dx, v = r * dy;
    pix[0] = c.fx * u + 0.0 * v + c.ppx;
    pix[1] = 0.0 * u + c.fy * v + c.ppy;
    return true;
}

// ---- R = V U^T of H (reflection fixed): Kabsch through the Jacobi eigen-decomposition of H^T H, see the oracle header ----
__device__ void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

template <int P, int Q>
__device__ __forceinline__ void jacobi_step(double *A, double *V)
{
    const double apq = A[3 * P + Q];
    if (apq == 0.0) return;
    const double theta = (A[3 * Q + Q] - A[3 * P + P]) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double akp = A[3 * k + P], akq = A[3 * k + Q];
        A[3 * k + P] = c * akp - s * akq; A[3 * k + Q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double apk = A[3 * P + k], aqk = A[3 * Q + k];
        A[3 * P + k] = c * apk - s * aqk; A[3 * Q + k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[3 * k + P], vkq = V[3 * k + Q];
        V[3 * k + P] = c * vkp - s * vkq; V[3 * k + Q] = s * vkp + c * vkq;
    }
}

__device__ void kabsch_rotation(const float *Hf, float *Rf)
{
    double H[9], A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = (double)Hf[k];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) A[3 * r + c] = H[r] * H[c] + H[3 + r] * H[3 + c] + H[6 + r] * H[6 + c];
    for (int sweep = 0; sweep < 8; ++sweep) { jacobi_step<0, 1>(A, V); jacobi_step<0, 2>(A, V); jacobi_step<1, 2>(A, V); }
    // the two largest eigenvalues, first maximum on ties; selects instead of dynamic indexing (registers, not scratch)
    const double e0 = A[0], e1 = A[4], e2 = A[8];
    int i0 = 0;
    if (e1 > e0) i0 = 1;
    if (e2 > (i0 == 0 ? e0 : e1)) i0 = 2;
    auto eig = [&](int i) { return i == 0 ? e0 : i == 1 ? e1 : e2; };
    int i1 = i0 == 0 ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) if (k != i0 && eig(k) > eig(i1)) i1 = k;
    auto col = [&](int i, int r) { return i == 0 ? V[3 * r] : i == 1 ? V[3 * r + 1] : V[3 * r + 2]; };
    const double v1[3] = {col(i0, 0), col(i0, 1), col(i0, 2)}, v2[3] = {col(i1, 0), col(i1, 1), col(i1, 2)};
    double v3[3], u1[3], u2[3], u3[3];
    const double l1 = eig(i0), l2 = eig(i1);
    const double s1 = sqrt(l1 > 0 ? l1 : 0.0), s2 = sqrt(l2 > 0 ? l2 : 0.0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        u1[r] = H[3 * r] * v1[0] + H[3 * r + 1] * v1[1] + H[3 * r + 2] * v1[2];
        u2[r] = H[3 * r] * v2[0] + H[3 * r + 1] * v2[1] + H[3 * r + 2] * v2[2];
    }
    if (s1 > 0.0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] /= s1;
    } else { u1[0] = 1; u1[1] = 0; u1[2] = 0; }
    if (s2 > 1e-12 * s1 && s2 > 0.0) {
        const double d = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] = u2[r] / s2 - (d / s2) * u1[r];
        const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    } else {
        int k0 = 0;
        if (fabs(u1[1]) < fabs(u1[0])) k0 = 1;
        if (fabs(u1[2]) < fabs(k0 == 0 ? u1[0] : u1[1])) k0 = 2;
        const double ax2[3] = {k0 == 0 ? 1.0 : 0.0, k0 == 1 ? 1.0 : 0.0, k0 == 2 ? 1.0 : 0.0};
        cross3(u1, ax2, u2);
        const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    }
    cross3(v1, v2, v3);
    cross3(u1, u2, u3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rf[3 * r + c] = (float)(v1[r] * u1[c] + v2[r] * u2[c] + v3[r] * u3[c]);
}

// rot_ransac.cpp:96-99,122-126: rotate the ray in binary32 (cv::Matx), project, compare with the tracked pixel
__device__ bool inlier(const float *R, const float *p1, float c2x, float c2y, const hv_camera_model &cam2, double thr)
{
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s = s + R[3 * r + k] * p1[k];
        q[r] = s;
    }
    const double ray[3] = {(double)q[0], (double)q[1], (double)q[2]};
    double pix[2];
    if (!ray_to_pixel(cam2, ray, pix)) return false;
    const float qx = (float)pix[0], qy = (float)pix[1];
    const double dx = (double)(c2x - qx), dy = (double)(c2y - qy);
    return dx * dx + dy * dy <= thr;
}

template <int RT, int NG = 1>
__global__ __launch_bounds__(RT) void rot_ransac_kernel(RansacArgs a)
{
    constexpr int HPG = HYP / NG;                 // hypotheses of this workgroup: k_lo .. k_lo + HPG - 1
    __shared__ float s_p1[MAX_PTS * 3], s_p2[MAX_PTS * 3], s_c2[MAX_PTS * 2];
    __shared__ float s_R[HYP * 9];
    __shared__ int s_count[HYP], s_valid[HYP];
    __shared__ unsigned char s_in[MAX_PTS];
    __shared__ float s_Rb[9];
    __shared__ int s_best[4];
    __shared__ unsigned short s_map[MAX_PTS];     // compacted index -> original feature number
    __shared__ int s_chunk[MAX_PTS / 64 + 1];
    const int set = NG > 1 ? blockIdx.y : blockIdx.x, tid = threadIdx.x;
    const int k_lo = NG > 1 ? (int)blockIdx.x * HPG : 0;
    const int n_all = min(max(a.n_points[set], 0), a.max_points);      // a device-supplied count never indexes past the set's arrays
    const float *c1 = a.c1 + (size_t)set * a.max_points * 2, *c2 = a.c2 + (size_t)set * a.max_points * 2;
    int *status = a.status + (size_t)set * a.max_points;
    // ---- "Pick the left camera features for which track was found" (ransac_pipeline.cpp:106-112), in feature order:
    // 64-feature chunks counted with a ballot, chunk offsets by a short serial scan ----
    int n = n_all;
    if (a.lk_status) {
        const uint8_t *ls = a.lk_status + (size_t)set * a.max_points;
        const int nchunk = (n_all + 63) / 64, wave = tid >> 6, lane = tid & 63;
        for (int ch = wave; ch < nchunk; ch += RT / 64) {
            const int i = ch * 64 + lane;
            const unsigned long long m = __ballot(i < n_all && ls[i] == a.lk_tracked);
            if (lane == 0) s_chunk[ch] = __popcll(m);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int ch = 0; ch < nchunk; ++ch) { const int cnt = s_chunk[ch]; s_chunk[ch] = acc; acc += cnt; }
            s_chunk[nchunk] = acc;
        }
        __syncthreads();
        n = s_chunk[nchunk];
        for (int ch = wave; ch < nchunk; ch += RT / 64) {
            const int i = ch * 64 + lane;
            const bool on = i < n_all && ls[i] == a.lk_tracked;
            const unsigned long long m = __ballot(on);
            if (on) s_map[s_chunk[ch] + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
        }
    } else {
        for (int i = tid; i < n_all; i += RT) s_map[i] = (unsigned short)i;
    }
    __syncthreads();
    if (n < 2) {                                                             // ransac_pipeline.cpp:209: nothing to fit
        if (tid == 0 && k_lo == 0) { a.summary[2 * set] = 0; a.summary[2 * set + 1] = 0; }
        return;
    }
    const double thr = (double)a.threshold_pow2;
    // ---- rays of both frames (rot_ransac.cpp:63-66; pixelToRay's success is not checked there either) ----
    for (int i = tid; i < n; i += RT) {
        const int src = s_map[i];
        double r[3];
        pixel_to_ray(a.cam1, (double)c1[2 * src], (double)c1[2 * src + 1], r);
        s_p1[3 * i] = (float)r[0]; s_p1[3 * i + 1] = (float)r[1]; s_p1[3 * i + 2] = (float)r[2];
        pixel_to_ray(a.cam2, (double)c2[2 * src], (double)c2[2 * src + 1], r);
        s_p2[3 * i] = (float)r[0]; s_p2[3 * i + 1] = (float)r[1]; s_p2[3 * i + 2] = (float)r[2];
        s_c2[2 * i] = c2[2 * src]; s_c2[2 * i + 1] = c2[2 * src + 1];
    }
    if (tid < HYP) s_count[tid] = 0;
    __syncthreads();
    // ---- hypothesis k: the rotation of its two pairs (:80-87) ----
    if (tid < HPG) {
        const int hk = k_lo + tid;
        int i1, i2;
        if (a.draws) {                                                       // rng() % n (rot_ransac.cpp:82-83), n known only here
            i1 = (int)(a.draws[((size_t)set * HYP + hk) * 2] % (uint32_t)n);
            i2 = (int)(a.draws[((size_t)set * HYP + hk) * 2 + 1] % (uint32_t)n);
        } else {
            i1 = a.pairs[((size_t)set * HYP + hk) * 2]; i2 = a.pairs[((size_t)set * HYP + hk) * 2 + 1];
        }
        // caller-supplied index pairs outside [0, n) make the hypothesis invalid instead of reading outside the set
        const int ok = i1 != i2 && (unsigned)i1 < (unsigned)n && (unsigned)i2 < (unsigned)n;
        s_valid[hk] = ok;
        if (ok) {
            float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R[9];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const float *p = s_p1 + 3 * (w ? i2 : i1), *q = s_p2 + 3 * (w ? i2 : i1);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { const float prod = p[r] * q[c]; H[3 * r + c] = H[3 * r + c] + prod; }
            }
            kabsch_rotation(H, R);
#pragma unroll
            for (int k = 0; k < 9; ++k) s_R[9 * hk + k] = R[k];
        }
    }
    __syncthreads();
    // ---- inlier counts of every hypothesis (:90-97): (point, hypothesis) pairs over the threads ----
    for (int w = tid; w < n * HPG; w += RT) {
        const int kk = w / n, i = w - kk * n, k = k_lo + kk;                  // consecutive lanes: consecutive points of one hypothesis
        if (s_valid[k] && inlier(s_R + 9 * k, s_p1 + 3 * i, s_c2[2 * i], s_c2[2 * i + 1], a.cam2, thr)) atomicAdd(&s_count[k], 1);
    }
    __syncthreads();
    if constexpr (NG > 1) {
        // this workgroup's hypotheses go to the set's record; the last workgroup to arrive collects all of them and carries on alone
        int *rec_i = reinterpret_cast<int *>(a.split + (size_t)set * SPLIT_REC_BYTES);
        float *rec_R = reinterpret_cast<float *>(rec_i + 2 * HYP + 4);
        if (tid < HPG) {
            rec_i[k_lo + tid] = s_count[k_lo + tid];
            rec_i[HYP + k_lo + tid] = s_valid[k_lo + tid];
        }
        for (int i = tid; i < 9 * HPG; i += RT) rec_R[9 * k_lo + i] = s_valid[k_lo + i / 9] ? s_R[9 * k_lo + i] : 0.0f;
        __threadfence();                                                      // (every writer: its stores are visible device-wide before the ticket)
        __syncthreads();
        if (tid == 0) s_best[3] = atomicAdd(&rec_i[2 * HYP], 1);
        __syncthreads();
        if (s_best[3] != NG - 1) return;
        __threadfence();
        if (tid < HYP) {
            s_count[tid] = __hip_atomic_load(&rec_i[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_valid[tid] = __hip_atomic_load(&rec_i[HYP + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int i = tid; i < 9 * HYP; i += RT) s_R[i] = __hip_atomic_load(&rec_R[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) rec_i[2 * HYP] = 0;                                     // the ticket counter is ready for the next launch
        __syncthreads();
    }
    // ---- the loop's bookkeeping (:99-104): first maximum, stop after the first hypothesis with every point an inlier ----
    if (tid == 0) {
        int best = 0, bk = -1, visited = HYP;
        for (int k = 0; k < HYP; ++k) {
            if (!s_valid[k]) continue;
            if (s_count[k] > best) { best = s_count[k]; bk = k; }
            if (s_count[k] == n) { visited = k + 1; break; }
        }
        s_best[0] = best; s_best[1] = bk; s_best[2] = visited;
        a.summary[2 * set] = best; a.summary[2 * set + 1] = visited;
        float R[9];
        if (bk >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = s_R[9 * bk + k];
        } else {                                                             // bestInds stays {0, 1} (:78)
            float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { const float prod = s_p1[3 * w + r] * s_p2[3 * w + c]; H[3 * r + c] = H[3 * r + c] + prod; }
            kabsch_rotation(H, R);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s_Rb[k] = R[k];
    }
    __syncthreads();
    // ---- refit on the inliers of the best hypothesis (:107-118) ----
    for (int i = tid; i < n; i += RT) s_in[i] = inlier(s_Rb, s_p1 + 3 * i, s_c2[2 * i], s_c2[2 * i + 1], a.cam2, thr) ? 1 : 0;
    __syncthreads();
    if (tid < 64) {                                                          // lane e < 9 sums entry e of H in index order (binary32)
        const int r = tid / 3, c = tid - 3 * r;
        float h = 0.0f;
        int cnt = 0;
        if (tid < 9)
            for (int i = 0; i < n; ++i)
                if (s_in[i]) { const float prod = s_p1[3 * i + r] * s_p2[3 * i + c]; h = h + prod; ++cnt; }
        float H[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) H[k] = __shfl(h, k);
        if (tid == 0 && cnt >= 2) {
            float R[9];
            kabsch_rotation(H, R);
#pragma unroll
            for (int k = 0; k < 9; ++k) s_Rb[k] = R[k];
        }
    }
    __syncthreads();
    // ---- final classification (:120-130) ----
    for (int i = tid; i < n; i += RT) status[s_map[i]] = inlier(s_Rb, s_p1 + 3 * i, s_c2[2 * i], s_c2[2 * i + 1], a.cam2, thr) ? 0 : 3;
    if (tid < 9) a.R[9 * (size_t)set + tid] = s_Rb[tid];
}

}  // namespace
}  // namespace hv

using hv::Ctx;

namespace hv {
int rot_ransac_alloc_split(Ctx *c)
{
    if (c->d_ransac_split) return HV_OK;
    const int sets = std::max(16, c->num_cus / RT_SPLIT_GROUPS);       // every set count the auto rule sends to the split form
    if (hipMalloc(reinterpret_cast<void **>(&c->d_ransac_split), SPLIT_REC_BYTES * (size_t)sets) != hipSuccess) return HV_ERR_NOMEM;
    HV_HIP(c, hipMemsetAsync(c->d_ransac_split, 0, SPLIT_REC_BYTES * (size_t)sets, c->stream));
    c->ransac_split_sets = sets;
    return HV_OK;
}
}  // namespace hv

extern "C" {

// CameraBase / PinholeCamera / FisheyeCamera constructors (camera.cpp:24-36,152-167,318-351): the derived fields
int hv_camera_model_init(hv_camera_model *m)
{
    if (!m || (m->kind != 0 && m->kind != 1) || m->n_coeffs < 0 || m->n_coeffs > 4) return HV_ERR_INVALID;
    const double K[9] = {m->fx, 0, m->ppx, 0, m->fy, m->ppy, 0, 0, 1};
    {   // 3x3 inverse through cofactors and one reciprocal of the determinant
        const double c00 = K[4] * K[8] - K[5] * K[7], c01 = K[5] * K[6] - K[3] * K[8], c02 = K[3] * K[7] - K[4] * K[6];
        const double det = K[0] * c00 + K[1] * c01 + K[2] * c02, id = 1.0 / det;
        double *o = m->kinv;
        o[0] = c00 * id; o[1] = (K[2] * K[7] - K[1] * K[8]) * id; o[2] = (K[1] * K[5] - K[2] * K[4]) * id;
        o[3] = c01 * id; o[4] = (K[0] * K[8] - K[2] * K[6]) * id; o[5] = (K[2] * K[3] - K[0] * K[5]) * id;
        o[6] = c02 * id; o[7] = (K[1] * K[6] - K[0] * K[7]) * id; o[8] = (K[0] * K[4] - K[1] * K[3]) * id;
    }
    m->n_table = 0; m->max_theta = m->max_r = 0;
    if (m->kind == 0) {
        m->distortion_enabled = !(m->n_coeffs == 0 || (m->n_coeffs == 1 && m->coeffs[0] == 0.)) ? 1 : 0;
        if (m->distortion_enabled && m->n_coeffs != 3) return HV_ERR_INVALID;                   // camera.cpp:163
        if (m->rotation_enabled) {
            double d = 0;
            for (int i = 0; i < 9; ++i) { const double e = m->rotation[i] - (i % 4 == 0 ? 1.0 : 0.0); d += e * e; }
            m->rotation_enabled = std::sqrt(d) > 1e-8 ? 1 : 0;                                  // isRotated, camera.cpp:125-127
        }
        return HV_OK;
    }
    m->rotation_enabled = 0;
    m->distortion_enabled = m->n_coeffs > 1 ? 1 : 0;
    if (m->distortion_enabled && m->n_coeffs != 4) return HV_ERR_INVALID;
    auto distort = [&](double theta, double *der) {
        if (!m->distortion_enabled) { if (der) *der = 1.0; return theta; }
        const double *k = m->coeffs, t = theta, t2 = t * t;
        if (der) *der = 1 + 3 * t2 * (k[0] + 5.0 / 3 * t2 * (k[1] + 7.0 / 5 * t2 * (k[2] + 9.0 / 7 * t2 * k[3])));
        return t * (1 + t2 * (k[0] + t2 * (k[1] + t2 * (k[2] + t2 * k[3]))));
    };
    m->max_theta = 0.5 * m->max_valid_fov_deg / 180.0 * M_PI;
    m->max_r = distort(m->max_theta, nullptr);
    if (m->distortion_enabled) {
        const double eps = 0.01 / ((m->fx + m->fy) * 0.5), step = m->max_r / 50.0;
        double theta = 0;
        for (int i = 0; i < 50; ++i) {
            const double r = (i + 0.5) * step;
            double th = theta, d, res = -1;
            for (int it = 0; it < 20; ++it) {
                const double dr = distort(th, &d) - r, dt = dr / d;
                th -= dt;
                if (std::fabs(dt) < eps) { res = th > 0.0 ? th : 0.0; break; }
            }
            if (res < 0) return HV_ERR_INVALID;                                                  // the reference asserts here
            m->table[m->n_table++] = res;
            theta = res + step;
        }
    }
    return HV_OK;
}

namespace hv { namespace {
// knob rot_ransac_threads (tests only): 256 / 1024 force the many-frames / the few-frames instantiation at any batch size, 25 the
// split form (RT_SPLIT_GROUPS workgroups per set); auto: split while every workgroup of the launch gets a CU of its own, 1024 threads
// up to 64 sets, 256 beyond
int launch_rot_ransac(Ctx *c, int n_sets, RansacArgs &a)
{
    const int rt_forced = c->knob.rot_ransac_threads;
    // the split form's records exist once per context (rot_ransac_alloc_split, hv_create): a launch with more sets than they hold --
    // only the forced knob can ask for that -- takes the 1024-thread form instead of allocating in the launch path (r05 advisor: a
    // synchronise / free / malloc here would invalidate a stream capture)
    const bool split = (rt_forced == RT_SPLIT_GROUPS || (rt_forced == 0 && n_sets * RT_SPLIT_GROUPS <= c->num_cus)) && n_sets <= c->ransac_split_sets;
    if (split) {
        a.split = c->d_ransac_split;
        hipLaunchKernelGGL((rot_ransac_kernel<RT_SPLIT, RT_SPLIT_GROUPS>), dim3(RT_SPLIT_GROUPS, (unsigned)n_sets), dim3(RT_SPLIT), 0, c->stream, a);
    } else if (rt_forced == RT_FEW || (rt_forced != RT_MANY && n_sets <= 64))
        hipLaunchKernelGGL((rot_ransac_kernel<RT_FEW>), dim3((unsigned)n_sets), dim3(RT_FEW), 0, c->stream, a);
    else
        hipLaunchKernelGGL((rot_ransac_kernel<RT_MANY>), dim3((unsigned)n_sets), dim3(RT_MANY), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}
} }

int hv_rot_ransac_batch_dev(hv_ctx *h, int n_sets, int max_points, const int *n_points_dev, const float *c1_dev, const float *c2_dev,
                            const hv_camera_model *cam1, const hv_camera_model *cam2, const int *pairs_dev, float threshold_pow2,
                            int *status_dev, float *R_dev, int *summary_dev)
{
    Ctx *c = hv::ctx_of(h);
    if (!c || n_sets < 0 || max_points < 2 || max_points > hv::MAX_PTS || !cam1 || !cam2) return HV_ERR_INVALID;
    if (n_sets > 0 && (!n_points_dev || !c1_dev || !c2_dev || !pairs_dev || !status_dev || !R_dev || !summary_dev)) return HV_ERR_INVALID;
    if (n_sets == 0) return HV_OK;
    hv::RansacArgs a{};
    a.max_points = max_points; a.n_points = n_points_dev; a.c1 = c1_dev; a.c2 = c2_dev; a.pairs = pairs_dev;
    a.threshold_pow2 = threshold_pow2; a.status = status_dev; a.R = R_dev; a.summary = summary_dev;
    a.cam1 = *cam1; a.cam2 = *cam2;
    hv::ScopedKernelTime tm(c, HV_K_ROT_RANSAC);
    return hv::launch_rot_ransac(c, n_sets, a);
}

int hv_rot_ransac_lk_batch_dev(hv_ctx *h, int n_sets, int max_points, const int *n_points_dev, const float *c1_dev, const float *c2_dev,
                               const uint8_t *lk_status_dev, int lk_tracked_value, const hv_camera_model *cam1,
                               const hv_camera_model *cam2, const uint32_t *draws_dev, float threshold_pow2, int *status_dev,
                               float *R_dev, int *summary_dev)
{
    Ctx *c = hv::ctx_of(h);
    if (!c || n_sets < 0 || max_points < 2 || max_points > hv::MAX_PTS || !cam1 || !cam2) return HV_ERR_INVALID;
    if (n_sets > 0 && (!n_points_dev || !c1_dev || !c2_dev || !lk_status_dev || !draws_dev || !status_dev || !R_dev || !summary_dev))
        return HV_ERR_INVALID;
    if (n_sets == 0) return HV_OK;
    hv::RansacArgs a{};
    a.max_points = max_points; a.n_points = n_points_dev; a.c1 = c1_dev; a.c2 = c2_dev; a.draws = draws_dev;
    a.lk_status = lk_status_dev; a.lk_tracked = lk_tracked_value;
    a.threshold_pow2 = threshold_pow2; a.status = status_dev; a.R = R_dev; a.summary = summary_dev;
    a.cam1 = *cam1; a.cam2 = *cam2;
    hv::ScopedKernelTime tm(c, HV_K_ROT_RANSAC);
    return hv::launch_rot_ransac(c, n_sets, a);
}

int hv_rot_ransac(hv_ctx *h, int n, const float *c1, const float *c2, const hv_camera_model *cam1, const hv_camera_model *cam2,
                  const int *pairs, float threshold_pow2, int *status, float *R, int *best_inlier_count, int *hypotheses_visited)
{
    Ctx *c = hv::ctx_of(h);
    if (!c || n < 2 || n > hv::MAX_PTS || !c1 || !c2 || !pairs || !status || !cam1 || !cam2) return HV_ERR_INVALID;
    const size_t o_c1 = 0, o_c2 = o_c1 + sizeof(float) * 2 * n, o_pairs = o_c2 + sizeof(float) * 2 * n;
    const size_t o_n = o_pairs + sizeof(int) * 2 * hv::HYP, o_st = o_n + 16, o_R = o_st + sizeof(int) * n, o_sum = o_R + 48;
    const size_t total = o_sum + 16;
    if (c->ransac_stage_bytes < total) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_ransac_stage) (void)hipFree(c->d_ransac_stage);
        c->d_ransac_stage = nullptr; c->ransac_stage_bytes = 0;
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_ransac_stage), total));
        c->ransac_stage_bytes = total;
    }
    unsigned char *d = c->d_ransac_stage;
    HV_HIP(c, hipMemcpyAsync(d + o_c1, c1, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_c2, c2, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_pairs, pairs, sizeof(int) * 2 * hv::HYP, hipMemcpyHostToDevice, c->stream));
    HV_HIP(c, hipMemcpyAsync(d + o_n, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    int rc = hv_rot_ransac_batch_dev(h, 1, n, reinterpret_cast<const int *>(d + o_n), reinterpret_cast<const float *>(d + o_c1),
                                     reinterpret_cast<const float *>(d + o_c2), cam1, cam2, reinterpret_cast<const int *>(d + o_pairs),
                                     threshold_pow2, reinterpret_cast<int *>(d + o_st), reinterpret_cast<float *>(d + o_R),
                                     reinterpret_cast<int *>(d + o_sum));
    if (rc != HV_OK) return rc;
    int summary[2] = {0, 0};
    HV_HIP(c, hipMemcpyAsync(status, d + o_st, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (R) HV_HIP(c, hipMemcpyAsync(R, d + o_R, sizeof(float) * 9, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipMemcpyAsync(summary, d + o_sum, sizeof(summary), hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    if (best_inlier_count) *best_inlier_count = summary[0];
    if (hypotheses_visited) *hypotheses_visited = summary[1];
    return HV_OK;
}

}  // extern "C"
