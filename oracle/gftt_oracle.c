/*
 * oracle/gftt_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product path (hybvio_amd/csrc, include/hybvio_hip.h) never
 * links, imports or calls anything in this directory.
 *
 * What it restates (the feature detector the
 * reference runs when it has no GPU image factory, i.e. the path `./main` takes
 * without -gpu; featureDetector "GPU-GFTT", codegen/parameter_definitions.c:311)
 * ----------------------------------------------------------------------------
 *   src/tracker/feature_detector.cpp:279-315  CpuCornerResponse:
 *        cv::cornerMinEigenVal(gray, response, gfttBlockSize, 3)
 *   src/tracker/feature_detector.cpp:393-417  CollectMax::cpuImplementation:
 *        arg-max of 16 * response per bs x bs block, bs = product of the reduce
 *        factors chosen from gfttMinDistance (:428-436), strict `>` against the
 *        running maximum and against gfttMinResponse, raster scan order; a key
 *        point is emitted for EVERY block (x = y = 0, response -1e10 when no
 *        pixel qualifies); the block count is floor(size / bs) because
 *        `std::ceil(response.height / bs)` divides integers (:395-396)
 *   src/tracker/feature_detector.cpp:618-634  detect(): std::stable_sort by
 *        descending response; `corners.resize(keypoints.size())` followed by
 *        push_back, which PREPENDS one (0, 0) point per key point (:629-631);
 *        applyMinDistance when maskRadius > 0
 *   src/tracker/feature_detector_legacy.cpp:177-213  applyMinDistance: greedy
 *        in-order filter against the previous corners and the corners kept so
 *        far, squared distance < r*r in float, stops at maxTracks
 *
 * cv::cornerMinEigenVal is OpenCV (modules/imgproc/src/corner.cpp,
 * cornerEigenValsVecs + calcMinEigenVal; Sobel via sepFilter2D, boxFilter). OpenCV
 * arrives through the empty submodule 3rdparty/mobile-cv-suite, so this file
 * restates the published 4.x algorithm:
 *   scale = 1 / (2^(ksize-1) * blockSize * 255)                      (8-bit input)
 *   Dx = Sobel(src, CV_32F, 1, 0, 3, scale): row kernel [-1 0 1], column kernel
 *        float(scale) * [1 2 1]   (cv::Sobel scales the smoothing kernel)
 *   Dy = Sobel(src, CV_32F, 0, 1, 3, scale): row kernel float(scale)*[1 2 1],
 *        column kernel [-1 0 1]
 *   cov = (Dx*Dx, Dx*Dy, Dy*Dy); boxFilter(cov, blockSize x blockSize, normalize
 *        = false); a = cov0 * 0.5f, b = cov1, c = cov2 * 0.5f;
 *   response = (a + c) - sqrt((a - c)*(a - c) + b*b);   borders BORDER_REFLECT_101.
 *
 * PARITY UNPINNED against real OpenCV: no reference test or fixture covers the
 * detector, and binary32 results of OpenCV's filters depend on its build (FMA in
 * the SIMD column filters) and, for the un-normalised box filter, on the history
 * of its running row / column sums (`s += in - out`). This restatement fixes one
 * order: symmetric 3-tap filters as k0*x0 + k1*(x-1 + x+1), the box as direct
 * sums ((p-1 + p0) + p+1) rows first then columns, no FMA contraction
 * (-ffp-contract=off). The HIP kernel evaluates exactly this sequence, so key
 * points and responses are bit-identical between the two.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* cv::cornerMinEigenVal(src, dst, blockSize, ksize = 3, BORDER_DEFAULT) for 8-bit input */
int orc_corner_min_eigen_val(const uint8_t *img, int w, int h, int stride, int block_size, float *response)
{
    if (!img || !response || w < 1 || h < 1 || block_size < 1 || (block_size & 1) == 0) return -1;
    const double scale_d = 1.0 / ((double)(1 << 2) * block_size * 255.0);
    const float k1 = (float)(1.0 * scale_d), k0 = (float)(2.0 * scale_d);     /* float(scale * [1 2 1]) */
    const size_t n = (size_t)w * h;
    float *dx = (float *)malloc(sizeof(float) * n), *dy = (float *)malloc(sizeof(float) * n);
    float *c0 = (float *)malloc(sizeof(float) * n), *c1 = (float *)malloc(sizeof(float) * n), *c2 = (float *)malloc(sizeof(float) * n);
    float *r0 = (float *)malloc(sizeof(float) * n), *r1 = (float *)malloc(sizeof(float) * n), *r2 = (float *)malloc(sizeof(float) * n);
    if (!dx || !dy || !c0 || !c1 || !c2 || !r0 || !r1 || !r2) { free(dx); free(dy); free(c0); free(c1); free(c2); free(r0); free(r1); free(r2); return -2; }
#define PIX(x, y) ((float)img[(size_t)reflect101((y), h) * stride + reflect101((x), w)])
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            /* Dx: rows [-1 0 1] (exact in float), columns k0*mid + k1*(top + bottom) */
            const float dt = PIX(x + 1, y - 1) - PIX(x - 1, y - 1);
            const float dm = PIX(x + 1, y) - PIX(x - 1, y);
            const float db = PIX(x + 1, y + 1) - PIX(x - 1, y + 1);
            const float vx = k0 * dm + k1 * (dt + db);
            /* Dy: rows k0*mid + k1*(left + right), columns [-1 0 1] */
            const float st = k0 * PIX(x, y - 1) + k1 * (PIX(x - 1, y - 1) + PIX(x + 1, y - 1));
            const float sb = k0 * PIX(x, y + 1) + k1 * (PIX(x - 1, y + 1) + PIX(x + 1, y + 1));
            const float vy = sb - st;
            dx[(size_t)y * w + x] = vx; dy[(size_t)y * w + x] = vy;
            c0[(size_t)y * w + x] = vx * vx; c1[(size_t)y * w + x] = vx * vy; c2[(size_t)y * w + x] = vy * vy;
        }
    }
#undef PIX
    const int hb = block_size / 2;
    /* un-normalised box filter: row sums, then column sums, left-to-right / top-to-bottom */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int i = -hb; i <= hb; i++) {
                const size_t p = (size_t)y * w + reflect101(x + i, w);
                if (i == -hb) { s0 = c0[p]; s1 = c1[p]; s2 = c2[p]; }
                else { s0 = s0 + c0[p]; s1 = s1 + c1[p]; s2 = s2 + c2[p]; }
            }
            r0[(size_t)y * w + x] = s0; r1[(size_t)y * w + x] = s1; r2[(size_t)y * w + x] = s2;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int j = -hb; j <= hb; j++) {
                const size_t p = (size_t)reflect101(y + j, h) * w + x;
                if (j == -hb) { s0 = r0[p]; s1 = r1[p]; s2 = r2[p]; }
                else { s0 = s0 + r0[p]; s1 = s1 + r1[p]; s2 = s2 + r2[p]; }
            }
            /* calcMinEigenVal */
            const float a = s0 * 0.5f, b = s1, c = s2 * 0.5f;
            const float amc = a - c;
            response[(size_t)y * w + x] = (a + c) - sqrtf(amc * amc + b * b);
        }
    }
    free(dx); free(dy); free(c0); free(c1); free(c2); free(r0); free(r1); free(r2);
    return 0;
}

/* CollectMax (feature_detector.cpp:428-436): block size from gfttMinDistance */
int orc_gftt_block_size(double gftt_min_distance)
{
    const int target = (int)gftt_min_distance;
    if (target >= 32) return 4 * 4 * 2;
    if (target >= 16) return 4 * 4;
    return 4 * 2;
}

/* CollectMax::cpuImplementation (feature_detector.cpp:393-417). kp: [nblocks][3] = x, y, response.
 * Returns the number of key points (= floor(w / bs) * floor(h / bs)). */
int orc_gftt_collect_max(const float *response, int w, int h, int bs, float min_response, float *kp)
{
    const int nby = h / bs, nbx = w / bs;             /* std::ceil(int / int): the ragged edge is skipped */
    int n = 0;
    for (int yb = 0; yb < nby; yb++) {
        for (int xb = 0; xb < nbx; xb++) {
            float max_response = -1e10f;
            int best_x = 0, best_y = 0;
            for (int y = yb * bs; y < (yb + 1) * bs && y < h; y++) {
                for (int x = xb * bs; x < (xb + 1) * bs && x < w; x++) {
                    const float r = (float)((double)response[(size_t)y * w + x] * 16.0);   /* CpuCornerResponse::GAIN */
                    if (r > max_response && r > min_response) { best_x = x; best_y = y; max_response = r; }
                }
            }
            kp[3 * n] = (float)best_x; kp[3 * n + 1] = (float)best_y; kp[3 * n + 2] = max_response;
            n++;
        }
    }
    return n;
}

/* FeatureDetector::applyMinDistance (feature_detector_legacy.cpp:177-213), in place; returns the new count */
int orc_apply_min_distance(float *corners, int n, const float *prev, int n_prev, int r, int max_tracks)
{
    int n_out = 0;
    const float r2 = (float)(r * r);
    for (int k = 0; k < n; k++) {
        const float cx = corners[2 * k], cy = corners[2 * k + 1];
        int near_other = 0;
        if (r > 0) {
            for (int i = 0; i < n_prev && !near_other; i++) {
                const float dx = prev[2 * i] - cx, dy = prev[2 * i + 1] - cy;
                if (dx * dx + dy * dy < r2) near_other = 1;
            }
            for (int i = 0; i < n_out && !near_other; i++) {
                const float dx = corners[2 * i] - cx, dy = corners[2 * i + 1] - cy;
                if (dx * dx + dy * dy < r2) near_other = 1;
            }
        }
        if (!near_other) { corners[2 * n_out] = cx; corners[2 * n_out + 1] = cy; n_out++; }
        if (n_out >= max_tracks) break;
    }
    return n_out;
}

/* FeatureDetectorImplementation::detect (feature_detector.cpp:610-634) on the CPU-fallback path.
 * corners: capacity 2 * nblocks points (x, y pairs). Returns the number of corners written. */
int orc_gftt_detect(const uint8_t *img, int w, int h, int stride, int block_size, double gftt_min_distance,
                    float min_response, const float *prev, int n_prev, int mask_radius, int max_tracks,
                    float *corners, int cap)
{
    const int bs = orc_gftt_block_size(gftt_min_distance);
    const int nk = (w / bs) * (h / bs);
    if (cap < 2 * nk) return -1;
    float *resp = (float *)malloc(sizeof(float) * (size_t)w * h);
    float *kp = (float *)malloc(sizeof(float) * 3 * (size_t)(nk > 0 ? nk : 1));
    int *order = (int *)malloc(sizeof(int) * (size_t)(nk > 0 ? nk : 1));
    if (!resp || !kp || !order) { free(resp); free(kp); free(order); return -2; }
    int rc = orc_corner_min_eigen_val(img, w, h, stride, block_size, resp);
    if (rc == 0) {
        orc_gftt_collect_max(resp, w, h, bs, min_response, kp);
        /* std::stable_sort, descending response: insertion sort is stable */
        for (int i = 0; i < nk; i++) {
            int j = i;
            while (j > 0 && kp[3 * order[j - 1] + 2] < kp[3 * i + 2]) { order[j] = order[j - 1]; j--; }
            order[j] = i;
        }
        /* corners.resize(n) then push_back: n zero points first (feature_detector.cpp:629-631) */
        for (int i = 0; i < nk; i++) { corners[2 * i] = 0.f; corners[2 * i + 1] = 0.f; }
        for (int i = 0; i < nk; i++) { corners[2 * (nk + i)] = kp[3 * order[i]]; corners[2 * (nk + i) + 1] = kp[3 * order[i] + 1]; }
        rc = 2 * nk;
        if (mask_radius > 0) rc = orc_apply_min_distance(corners, 2 * nk, prev, n_prev, mask_radius, max_tracks);
    }
    free(resp); free(kp); free(order);
    return rc;
}
