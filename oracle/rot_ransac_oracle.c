/*
 * CPU oracle for SURVEY.md section 8(f) row f4: the 2-point rotation RANSAC that runs on the LK output of every frame.
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file; the product path (hybvio_amd/csrc, hybvio_amd/host) never does.
 *
 * Restated from the reference:
 *   RotRansac::fit              src/tracker/rot_ransac.cpp:41-120  (100 hypotheses from rng() % n pairs, inlier test through
 *                               camera2.rayToPixel, best = first maximum, early exit when every point is an inlier, refit on
 *                               the inliers of the best hypothesis, final TRACKED / RANSAC_OUTLIER classification)
 *   withinInlierThreshold       rot_ransac.cpp:122-126
 *   solveRotation               rot_ransac.cpp:132-160  (H = sum p1 p2^T in binary32, R = V U^T of its SVD, reflection fixed)
 *   doRansac2 / threshold       src/tracker/ransac_pipeline.cpp:91-93,197-216
 *   std::mt19937                the C++ standard generator (seed ransacRngSeed = 4649, parameter_definitions.c:305)
 *   camera models               oracle/ingest_oracle.c (camera.cpp)
 *
 * PARITY UNPINNED: the reference has no test for this class, and the SVD is cv::SVD (OpenCV's binary32 Jacobi SVD, absent
 * here: SURVEY.md 8(c)). The SVD only enters through R = V U^T, which is defined here -- and identically in the kernel -- as
 * the Kabsch solution evaluated in binary64: eigen-decomposition of H^T H by cyclic Jacobi (fixed 8 sweeps over (0,1) (0,2)
 * (1,2), only + - * / sqrt, so that gcc and the GPU produce the same bits), u_k = H v_k / s_k for the two largest singular
 * values, third axes by cross products (which is the reference's det < 0 reflection fix), cast to binary32 like cv::Matx33f.
 * For a rank-deficient H (fewer than two independent pairs) cv::SVD returns an arbitrary completion; here a fixed one.
 * std::mt19937 is checked against its standard known answer (10000th output of the default seed = 4123659995).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_camera orc_camera;
int orc_camera_pixel_to_ray(const orc_camera *c, double px, double py, double *ray);
int orc_camera_ray_to_pixel(const orc_camera *c, const double *ray0, double *pix);

/* ---- std::mt19937 ---- */
typedef struct orc_mt19937 { uint32_t mt[624]; int idx; } orc_mt19937;

void orc_mt19937_seed(orc_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t orc_mt19937_next(orc_mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* convenience for the tests: n raw outputs after seeding */
void orc_mt19937_draws(uint32_t seed, int skip, int n, uint32_t *out)
{
    orc_mt19937 g;
    orc_mt19937_seed(&g, seed);
    for (int i = 0; i < skip; ++i) (void)orc_mt19937_next(&g);
    for (int i = 0; i < n; ++i) out[i] = orc_mt19937_next(&g);
}

/* ---- solveRotation (rot_ransac.cpp:132-160) ---- */
static void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

/* R = V U^T (reflection fixed) of the 3x3 H (row-major binary32), see the header for the definition */
void orc_kabsch_rotation(const float *Hf, float *Rf)
{
    double H[9], A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) H[k] = (double)Hf[k];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[3 * r + c] = H[r] * H[c] + H[3 + r] * H[3 + c] + H[6 + r] * H[6 + c];   /* H^T H */
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int e = 0; e < 3; ++e) {
            const int p = PQ[e][0], q = PQ[e][1];
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; ++k) {                      /* A <- A J (columns p, q) */
                const double akp = A[3 * k + p], akq = A[3 * k + q];
                A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
            }
            for (int k = 0; k < 3; ++k) {                      /* A <- J^T A (rows p, q) */
                const double apk = A[3 * p + k], aqk = A[3 * q + k];
                A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 3; ++k) {                      /* V <- V J */
                const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
            }
        }
    /* the two largest eigenvalues (first maximum on ties) */
    int i0 = 0;
    if (A[4] > A[3 * i0 + i0]) i0 = 1;
    if (A[8] > A[3 * i0 + i0]) i0 = 2;
    int i1 = i0 == 0 ? 1 : 0;
    for (int k = 0; k < 3; ++k) if (k != i0 && A[3 * k + k] > A[3 * i1 + i1]) i1 = k;
    double v1[3] = {V[i0], V[3 + i0], V[6 + i0]}, v2[3] = {V[i1], V[3 + i1], V[6 + i1]}, v3[3], u1[3], u2[3], u3[3];
    const double s1 = sqrt(A[3 * i0 + i0] > 0 ? A[3 * i0 + i0] : 0.0), s2 = sqrt(A[3 * i1 + i1] > 0 ? A[3 * i1 + i1] : 0.0);
    for (int r = 0; r < 3; ++r) {
        u1[r] = H[3 * r] * v1[0] + H[3 * r + 1] * v1[1] + H[3 * r + 2] * v1[2];
        u2[r] = H[3 * r] * v2[0] + H[3 * r + 1] * v2[1] + H[3 * r + 2] * v2[2];
    }
    if (s1 > 0.0) for (int r = 0; r < 3; ++r) u1[r] /= s1;
    else { u1[0] = 1; u1[1] = 0; u1[2] = 0; }
    if (s2 > 1e-12 * s1 && s2 > 0.0) {
        /* re-orthogonalise against u1 (exact for an exact SVD, guards the last bits) */
        const double d = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
        for (int r = 0; r < 3; ++r) u2[r] = u2[r] / s2 - (d / s2) * u1[r];
        const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    } else {                                                    /* rank <= 1: a fixed perpendicular completion */
        int k0 = 0;                                             /* the coordinate axis least aligned with u1 */
        if (fabs(u1[1]) < fabs(u1[k0])) k0 = 1;
        if (fabs(u1[2]) < fabs(u1[k0])) k0 = 2;
        const double ax2[3] = {k0 == 0 ? 1.0 : 0.0, k0 == 1 ? 1.0 : 0.0, k0 == 2 ? 1.0 : 0.0};
        cross3(u1, ax2, u2);
        const double nn = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    }
    cross3(v1, v2, v3);
    cross3(u1, u2, u3);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rf[3 * r + c] = (float)(v1[r] * u1[c] + v2[r] * u2[c] + v3[r] * u3[c]);
}

/* H = sum_{j in inds} p1[j] p2[j]^T accumulated in binary32 in index order (cv::Matx arithmetic), then the rotation */
void orc_solve_rotation(const float *p1, const float *p2, const int *inds, int n_inds, float *R)
{
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_inds; ++i) {
        const float *a = p1 + 3 * inds[i], *b = p2 + 3 * inds[i];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { const float prod = a[r] * b[c]; H[3 * r + c] = H[3 * r + c] + prod; }
    }
    orc_kabsch_rotation(H, R);
}

static int inlier(const float *R, const float *p1, const float *c2, const orc_camera *cam2, double threshold_pow2)
{
    float q[3];
    for (int r = 0; r < 3; ++r) {                               /* cv::Matx33f * cv::Matx31f: binary32, k = 0, 1, 2 */
        float s = 0.0f;
        for (int k = 0; k < 3; ++k) s = s + R[3 * r + k] * p1[k];
        q[r] = s;
    }
    const double ray[3] = {(double)q[0], (double)q[1], (double)q[2]};
    double pix[2];
    if (!orc_camera_ray_to_pixel(cam2, ray, pix)) return 0;
    const float qx = (float)pix[0], qy = (float)pix[1];
    const double dx = (double)(c2[0] - qx), dy = (double)(c2[1] - qy);          /* float difference, then double (rot_ransac.cpp:123) */
    return dx * dx + dy * dy <= threshold_pow2;
}

/* RotRansac::fit. c1, c2: [n][2] pixels; draws: the raw outputs the generator would produce from here on (at least 200);
 * status[i] = 0 TRACKED / 3 RANSAC_OUTLIER (track.hpp:9-21); returns the number of draws consumed. */
int orc_rot_ransac_fit(const float *c1, const float *c2, int n, const orc_camera *cam1, const orc_camera *cam2,
                       const uint32_t *draws, float threshold_pow2, int *status, float *R_out, int *best_inlier_count)
{
    float *p1 = (float *)malloc(sizeof(float) * 3 * (size_t)n), *p2 = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    int *inl = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double r[3];
        orc_camera_pixel_to_ray(cam1, (double)c1[2 * i], (double)c1[2 * i + 1], r);      /* "Does not check success." */
        for (int k = 0; k < 3; ++k) p1[3 * i + k] = (float)r[k];
        orc_camera_pixel_to_ray(cam2, (double)c2[2 * i], (double)c2[2 * i + 1], r);
        for (int k = 0; k < 3; ++k) p2[3 * i + k] = (float)r[k];
    }
    int best_inds[2] = {0, 1}, best = 0, used = 0;
    const double thr = (double)threshold_pow2;
    float R[9];
    for (int k = 0; k < 100; ++k) {
        const int ind1 = (int)(draws[used] % (uint32_t)n), ind2 = (int)(draws[used + 1] % (uint32_t)n);
        used += 2;
        if (ind1 == ind2) continue;
        const int inds[2] = {ind1, ind2};
        orc_solve_rotation(p1, p2, inds, 2, R);
        int count = 0;
        for (int i = 0; i < n; ++i) count += inlier(R, p1 + 3 * i, c2 + 2 * i, cam2, thr);
        if (count > best) { best = count; best_inds[0] = ind1; best_inds[1] = ind2; }
        if (count == n) break;
    }
    *best_inlier_count = best;
    orc_solve_rotation(p1, p2, best_inds, 2, R);
    int ni = 0;
    for (int i = 0; i < n; ++i) if (inlier(R, p1 + 3 * i, c2 + 2 * i, cam2, thr)) inl[ni++] = i;
    if (ni >= 2) orc_solve_rotation(p1, p2, inl, ni, R);
    for (int i = 0; i < n; ++i) status[i] = inlier(R, p1 + 3 * i, c2 + 2 * i, cam2, thr) ? 0 : 3;
    memcpy(R_out, R, sizeof R);
    free(p1); free(p2); free(inl);
    return used;
}
