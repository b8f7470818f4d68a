/*
 * CPU oracle for SURVEY.md section 8(f) row f2: image ingest (colour -> gray copy, undistort / rectify remap).
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file; the product path (hybvio_amd/csrc, hybvio_amd/host) never does.
 *
 * PARITY UNPINNED for the image operations: the reference has no test or golden vector for the remap or the gray copy,
 * and its own build cannot be compiled here (OpenCV, Eigen and accelerated-arrays are absent: SURVEY.md 8(c)). The CAMERA
 * MODELS are pinned by the reference's own tests (test/camera.cpp:7-168: the Matlab pinhole projection, the distorted
 * pinhole ray <-> pixel pair, the fisheye cases), re-expressed in tests/test_oracle_ingest.py. What is restated:
 *
 *   remap            src/tracker/undistorter.cpp:71-110   the CPU branch of UndistorterImplementation::undistort
 *                    (per rectified pixel: pixelToRay of the rectified camera, rayToPixel of the original camera,
 *                    range test, bilinear taps accumulated in float in the order (0,0) (0,1) (1,0) (1,1),
 *                    int(out + 0.5) evaluated in double)
 *   camera models    src/tracker/camera.cpp:93-221 (pinhole, radial k1..k3, Newton undistort, optional rotation),
 *                    camera.cpp:264-381 (fisheye, k1..k4, table-started Newton), camera.cpp:24-36 (camera matrix),
 *                    undistorter.cpp:150-168 (buildMono: the rectified pinhole of the mono case)
 *   colour -> gray   src/tracker/image.cpp:351-367: coefficients {0.299, 0.587, 0.114[, 0]} applied to channels
 *                    0,1,2 by accelerated-arrays' pixelwiseAffine. That operation lives in the absent third-party
 *                    library; its rounding is not visible from the reference tree. Defined here (and in the
 *                    kernel) as  (uint8) (int) (((0.299f*c0 + 0.587f*c1) + 0.114f*c2) + 0.5f)  in float32 with
 *                    no contraction -- an assumption, stated as such in DESIGN.md.
 *
 * One reference behaviour needs a definition: undistorter.cpp:99 reads input.at(y0 + iy, x0 + ix) without a
 * bounds check, so a source position in the last column / row reaches one element past the row / the image.
 * On the continuous cv::Mat the reference uses that is the flat-memory neighbour; past the end of the buffer it
 * is undefined. Oracle and kernel both use: tap = flat[(y0+iy)*w + (x0+ix)] if that index is < w*h, else 0.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_camera {
    int kind;               /* 0 pinhole, 1 fisheye */
    double K[9], Kinv[9];   /* row-major camera matrix and its inverse (camera.cpp:24-36) */
    int ncoeff;
    double coeff[4];
    int distortion;
    int rotation_enabled;
    double R[9];            /* row-major */
    double max_theta, max_r;/* fisheye */
    double table[50];
    int ntable;
} orc_camera;

static double cam_focal(const orc_camera *c) { return (c->K[0] + c->K[4]) * 0.5; }

/* 3x3 inverse through cofactors and one reciprocal of the determinant (what Eigen's fixed-size inverse does) */
static void inv3(const double *m, double *o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02, id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* ---- fisheye (camera.cpp:264-381) ---- */
static double fish_distort(const orc_camera *c, double theta, double *der)
{
    if (!c->distortion) { if (der) *der = 1.0; return theta; }
    const double *k = c->coeff, t = theta, t2 = t * t;
    if (der) *der = 1 + 3 * t2 * (k[0] + 5.0 / 3 * t2 * (k[1] + 7.0 / 5 * t2 * (k[2] + 9.0 / 7 * t2 * k[3])));
    return t * (1 + t2 * (k[0] + t2 * (k[1] + t2 * (k[2] + t2 * k[3]))));
}

static double fish_newton(const orc_camera *c, double r, double theta0)
{
    const double eps = 0.01 / cam_focal(c);
    double theta = theta0, d;
    for (int it = 0; it < 20; ++it) {
        const double dr = fish_distort(c, theta, &d) - r, dt = dr / d;
        theta -= dt;
        if (fabs(dt) < eps) return theta > 0.0 ? theta : 0.0;
    }
    return -1;
}

static double fish_undistort(const orc_camera *c, double r)
{
    const size_t n = (size_t)c->ntable;
    double f = r / c->max_r; if (!(f > 0.0)) f = 0.0;
    size_t i = (size_t)(f * (double)n); if (i > n - 1) i = n - 1;
    const double th = fish_newton(c, r, c->table[i]);
    return th < 0 ? r : th;
}

/* kind, fx fy ppx ppy, ncoeff coefficients, optional rotation (pinhole only), valid field of view (fisheye) */
orc_camera *orc_camera_create(int kind, double fx, double fy, double ppx, double ppy, int ncoeff, const double *coeff,
                              const double *rotation_rowmajor, double max_valid_fov_deg)
{
    orc_camera *c = (orc_camera *)calloc(1, sizeof(orc_camera));
    c->kind = kind;
    const double K[9] = {fx, 0, ppx, 0, fy, ppy, 0, 0, 1};
    memcpy(c->K, K, sizeof K);
    inv3(c->K, c->Kinv);
    c->ncoeff = ncoeff;
    for (int i = 0; i < ncoeff && i < 4; ++i) c->coeff[i] = coeff[i];
    if (kind == 0) {
        c->distortion = !(ncoeff == 0 || (ncoeff == 1 && coeff[0] == 0.));          /* camera.cpp:160-166 */
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(c->R, I, sizeof I);
        if (rotation_rowmajor) {
            double d = 0;
            for (int i = 0; i < 9; ++i) d += (rotation_rowmajor[i] - I[i]) * (rotation_rowmajor[i] - I[i]);
            if (sqrt(d) > 1e-8) { c->rotation_enabled = 1; memcpy(c->R, rotation_rowmajor, sizeof I); }
        }
    } else {
        c->distortion = ncoeff > 1;                                                    /* camera.cpp:326 */
        c->max_theta = 0.5 * max_valid_fov_deg / 180.0 * M_PI;
        c->max_r = fish_distort(c, c->max_theta, NULL);
        if (c->distortion) {                                                            /* camera.cpp:339-350 */
            double theta = 0;
            const double step = c->max_r / 50.0;
            for (int i = 0; i < 50; ++i) {
                theta = fish_newton(c, (i + 0.5) * step, theta);
                c->table[c->ntable++] = theta;
                theta += step;
            }
        }
    }
    return c;
}

void orc_camera_destroy(orc_camera *c) { free(c); }

/* ---- pinhole (camera.cpp:93-221) ---- */
static void pin_distort(const orc_camera *c, double *x, double *y, double *J)
{
    if (!c->distortion) { J[0] = 1; J[1] = 0; J[2] = 0; J[3] = 1; return; }
    const double *k = c->coeff, X = *x, Y = *y, r2 = X * X + Y * Y;
    const double theta = 1 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]));
    const double dth = k[0] + r2 * (k[1] * 2 + r2 * k[2] * 3);
    J[0] = theta + X * dth * 2 * X; J[1] = X * dth * 2 * Y;
    J[2] = Y * dth * 2 * X;         J[3] = theta + Y * dth * 2 * Y;
    *x = X * theta; *y = Y * theta;
}

static void pin_undistort(const orc_camera *c, double *px, double *py)
{
    if (!c->distortion) return;
    const double dx = *px, dy = *py;
    double x = dx, y = dy, nrm;
    int it = 0;
    do {
        double qx = x, qy = y, J[4];
        pin_distort(c, &qx, &qy, J);
        /* 2x2 inverse as Eigen computes it: adjugate times 1/det */
        const double id = 1.0 / (J[0] * J[3] - J[1] * J[2]);
        const double ex = dx - qx, ey = dy - qy;
        const double sx = (J[3] * id) * ex + (-J[1] * id) * ey, sy = (-J[2] * id) * ex + (J[0] * id) * ey;
        x += sx; y += sy;
        nrm = sqrt(sx * sx + sy * sy);
    } while (nrm > 1e-5 && ++it < 100);
    *px = x; *py = y;
}

int orc_camera_pixel_to_ray(const orc_camera *c, double px, double py, double *ray)
{
    if (c->kind == 0) {
        double x = (px - c->K[2]) / c->K[0], y = (py - c->K[5]) / c->K[4];
        pin_undistort(c, &x, &y);
        const double n = sqrt(x * x + y * y + 1.0);
        double r[3] = {x / n, y / n, 1.0 / n};
        if (c->rotation_enabled) {
            const double *R = c->R;
            const double t[3] = {R[0] * r[0] + R[1] * r[1] + R[2] * r[2], R[3] * r[0] + R[4] * r[1] + R[5] * r[2],
                                 R[6] * r[0] + R[7] * r[1] + R[8] * r[2]};
            memcpy(r, t, sizeof t);
        }
        memcpy(ray, r, sizeof r);
        return 1;
    }
    const double *Ki = c->Kinv;
    const double u = Ki[0] * px + Ki[1] * py + Ki[2], v = Ki[3] * px + Ki[4] * py + Ki[5];
    const double r = sqrt(u * u + v * v), dxn = u / r, dyn = v / r;
    int ok = 1;
    double theta = r;
    if (r > c->max_r) { theta = c->max_theta; ok = 0; }
    else if (c->distortion) theta = fish_undistort(c, r);
    const double s = sin(theta);
    ray[0] = s * dxn; ray[1] = s * dyn; ray[2] = cos(theta);
    return ok;
}

int orc_camera_ray_to_pixel(const orc_camera *c, const double *ray0, double *pix)
{
    if (c->kind == 0) {
        double r[3] = {ray0[0], ray0[1], ray0[2]};
        if (c->rotation_enabled) {
            const double *R = c->R;   /* R^T * ray */
            const double t[3] = {R[0] * r[0] + R[3] * r[1] + R[6] * r[2], R[1] * r[0] + R[4] * r[1] + R[7] * r[2],
                                 R[2] * r[0] + R[5] * r[1] + R[8] * r[2]};
            memcpy(r, t, sizeof t);
        }
        if (r[2] <= 0) return 0;
        const double iz = 1.0 / r[2];
        double x = r[0] * iz, y = r[1] * iz, J[4];
        pin_distort(c, &x, &y, J);
        pix[0] = c->K[0] * x + c->K[1] * y + c->K[2] * (r[2] * iz);
        pix[1] = c->K[3] * x + c->K[4] * y + c->K[5] * (r[2] * iz);
        return 1;
    }
    if (ray0[2] <= 0) return 0;
    const double inv = 1.0 / sqrt(ray0[0] * ray0[0] + ray0[1] * ray0[1] + ray0[2] * ray0[2]);
    const double theta = acos(ray0[2] * inv);
    if (theta > c->max_theta) return 0;
    const double r = fish_distort(c, theta, NULL);
    const double n2 = ray0[0] * ray0[0] + ray0[1] * ray0[1];
    double dx = ray0[0], dy = ray0[1];
    if (n2 > 0) { const double n = sqrt(n2); dx /= n; dy /= n; }     /* Eigen normalized(): untouched when the norm is 0 */
    const double u = r * dx, v = r * dy;
    pix[0] = c->K[0] * u + c->K[1] * v + c->K[2];
    pix[1] = c->K[3] * u + c->K[4] * v + c->K[5];
    return 1;
}

/* undistorter.cpp:56-60,86-90: rectified pixel -> position in the original image; valid = both camera calls succeeded */
void orc_undistort_map(const orc_camera *rect, const orc_camera *orig, int w, int h, double *pix_orig, uint8_t *valid)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double ray[3], p[2] = {0, 0};
            int ok = orc_camera_pixel_to_ray(rect, (double)x, (double)y, ray);
            if (ok) ok = orc_camera_ray_to_pixel(orig, ray, p);
            pix_orig[2 * ((size_t)y * w + x)] = p[0];
            pix_orig[2 * ((size_t)y * w + x) + 1] = p[1];
            valid[(size_t)y * w + x] = (uint8_t)ok;
        }
}

/* undistorter.cpp:84-110 with INTERPOLATE = true; in is a continuous w x h gray image */
void orc_undistort_apply(const uint8_t *in, int w, int h, const double *pix_orig, const uint8_t *valid, uint8_t *out)
{
    const long long total = (long long)w * h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double px = pix_orig[2 * ((size_t)y * w + x)], py = pix_orig[2 * ((size_t)y * w + x) + 1];
            float o = 0;
            if (valid[(size_t)y * w + x] && px >= 0 && px < w && py >= 0 && py < h) {
                const int x0 = (int)floor(px), y0 = (int)floor(py);
                const float xf = (float)(px - x0), yf = (float)(py - y0);
                for (int iy = 0; iy < 2; ++iy) {
                    const float wy = iy > 0 ? yf : (1 - yf);
                    for (int ix = 0; ix < 2; ++ix) {
                        const float wx = ix > 0 ? xf : (1 - xf);
                        const long long idx = (long long)(y0 + iy) * w + (x0 + ix);
                        const float tap = idx < total ? (float)in[idx] : 0.0f;
                        o += tap * wx * wy;
                    }
                }
            }
            out[(size_t)y * w + x] = (uint8_t)(int)((double)o + 0.5);
        }
}

/* image.cpp:351-367 (see the header for the rounding assumption); channels 3 or 4, interleaved, row stride in bytes */
void orc_color_to_gray(const uint8_t *in, int stride, int w, int h, int channels, uint8_t *out)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t *p = in + (size_t)y * stride + (size_t)x * channels;
            const float g = ((0.299f * (float)p[0] + 0.587f * (float)p[1]) + 0.114f * (float)p[2]) + 0.5f;
            out[(size_t)y * w + x] = (uint8_t)(int)g;
        }
}
