/*
 * oracle/pyrlk_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product path (hybvio_amd/csrc, include/hybvio_hip.h) never
 * links, imports or calls anything in this directory.
 *
 * What it restates
 * ----------------
 * HybVIO delegates the pixel arithmetic of its tracker hot path to OpenCV:
 *   src/tracker/image_pyramid.cpp:42-46   cv::buildOpticalFlowPyramid(img, pyr, Size(31,31), 3)
 *   src/tracker/optical_flow.cpp:46-49    cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, ...)
 *   src/tracker/optical_flow.cpp:10-59    status mapping + FLOW_OUT_OF_RANGE override
 * OpenCV arrives through the git submodule 3rdparty/mobile-cv-suite (.gitmodules:1-3),
 * which is EMPTY in the reference snapshot and whose pinned commit is not
 * recoverable. This file therefore restates the *published* OpenCV 4.x algorithm
 * (modules/video/src/lkpyramid.cpp: buildOpticalFlowPyramid, calcScharrDeriv /
 * ScharrDerivInvoker, LKTrackerInvoker; modules/imgproc/src/pyramids.cpp:
 * pyrDown_<FixPtCast<uchar,8>>; copyMakeBorder / borderInterpolate) in scalar C.
 *
 * PARITY UNPINNED: the reference holds no test, golden vector or fixture for the
 * pyramid or the LK tracker and neither OpenCV nor the
 * reference can be built or imported in this environment. The integer parts
 * (pyramid levels, Scharr gradients, fixed-point patch interpolation) are exact
 * restatements; see "accumulator note" for the one place where x86 OpenCV is
 * order-dependent.
 *
 * Accumulator note (A11/A12/A22, b1/b2)
 * -------------------------------------
 * lkpyramid.cpp selects `typedef int64 acctype; typedef int itemtype;` on
 * `__arm__ && !CV_NEON` and `typedef float acctype/itemtype` elsewhere, where the
 * SIMD code additionally sums in 4/8 float lanes. The float variant depends on
 * lane order at the 1e-7 relative level. This oracle (and the HIP kernel it
 * checks) uses the int64-accumulator variant: sums of the integer products are
 * exact and therefore independent of reduction order, which is what makes
 * status / IDs bit-reproducible between the CPU oracle and a 64-lane GPU
 * reduction. All remaining float arithmetic is a per-point scalar sequence that
 * is executed identically (IEEE-754 binary32, no FMA contraction, round-half-even
 * conversions) on both sides. Build with -ffp-contract=off.
 *
 * Accumulator MODES (r02): the float-accumulator variants OpenCV compiles off-ARM are
 * restated as well (orc_klt_track_mode), from the published lkpyramid.cpp, so that the distance
 * between what an x86 OpenCV build computes and the exact-integer variant the kernel implements can
 * be MEASURED (scripts/lk_accumulator_study.py, tests/test_oracle_lk_accumulators.py):
 *   ORC_ACC_INT64        exact integer sums (`__arm__ && !CV_NEON` typedefs)   -- the kernel's mode
 *   ORC_ACC_F32_SCALAR   `float acctype`, row-major scalar loop (a build without SIMD)
 *   ORC_ACC_F32_SIMD128  `CV_SIMD128 && !CV_NEON` universal-intrinsic code of OpenCV >= 4.1: A in 8-pixel steps as two
 *                        4-lane v_muladd groups (lane = pixel mod 4), b in 8-pixel steps through v_dotprod pairs
 *                        (pixel i with pixel i+4, exact int32) converted to f32 and added into two 4-lane vectors,
 *                        the 31 - 24 = 7 tail columns of every row in the scalar float accumulator, v_reduce_sum =
 *                        (a0 + a2) + (a1 + a3)
 *   ORC_ACC_F32_SIMD128_FMA  the same with v_muladd fused (CV_FMA3 builds: -mfma / AVX2 baseline)
 *   ORC_ACC_F32_SSE2_LEGACY  the hand-written `#if CV_SSE2` code of OpenCV 2.4 .. 4.0: A in 4-pixel steps
 *                        (_mm_add_ps(q, _mm_mul_ps(f, f))), 3 tail columns, buf[0] + buf[1] + buf[2] + buf[3]; b as above
 *   ORC_ACC_F32_WIDE8    NOT an OpenCV code path (LKTrackerInvoker is written with 128-bit types and stays 4-lane in
 *                        AVX2 builds): an 8-lane order, included as a sensitivity probe only
 * The CV_NEON block (ARM builds) is not restated. The modes differ ONLY in the five sums; everything else is shared.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_MAX_LEVELS 8

typedef struct {
    int w, h, pad;       /* interior size and border width (= LK window size)    */
    int gstride;         /* elements per padded gray row  = w + 2*pad            */
    int dstride;         /* int16 elements per padded deriv row = 2*(w + 2*pad)  */
    uint8_t *gray;       /* padded, BORDER_REFLECT_101                           */
    int16_t *deriv;      /* padded, BORDER_CONSTANT(0), interleaved [dx,dy]      */
} orc_level;

typedef struct {
    int nlevels;         /* maxLevel + 1 actually built */
    orc_level lv[ORC_MAX_LEVELS];
} orc_pyramid;

/* OpenCV borderInterpolate(p, len, BORDER_REFLECT_101) */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

static inline uint8_t *gray_at(const orc_level *L, int x, int y)
{
    return L->gray + (size_t)(y + L->pad) * L->gstride + (x + L->pad);
}
static inline int16_t *deriv_at(const orc_level *L, int x, int y)
{
    return L->deriv + (size_t)(y + L->pad) * L->dstride + 2 * (x + L->pad);
}

static int level_alloc(orc_level *L, int w, int h, int pad)
{
    L->w = w; L->h = h; L->pad = pad;
    L->gstride = w + 2 * pad;
    L->dstride = 2 * (w + 2 * pad);
    L->gray = (uint8_t *)malloc((size_t)L->gstride * (h + 2 * pad));
    L->deriv = (int16_t *)calloc((size_t)L->dstride * (h + 2 * pad), sizeof(int16_t));
    return (L->gray && L->deriv) ? 0 : -1;
}

/* copyMakeBorder(interior, padded, pad.., BORDER_REFLECT_101 | BORDER_ISOLATED) */
static void gray_make_border(orc_level *L)
{
    const int w = L->w, h = L->h, p = L->pad;
    for (int y = -p; y < h + p; y++) {
        const int sy = reflect101(y, h);
        for (int x = -p; x < w + p; x++) {
            if (x >= 0 && x < w && y >= 0 && y < h) continue;
            *gray_at(L, x, y) = *gray_at(L, reflect101(x, w), sy);
        }
    }
}

/* imgproc pyrDown_<FixPtCast<uchar,8>>: separable [1 4 6 4 1], (v + 128) >> 8,
 * dst = ((w+1)/2, (h+1)/2), source border REFLECT_101 on the level's own ROI. */
static void pyr_down(const orc_level *S, orc_level *D)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    const int sw = S->w, sh = S->h;
#pragma omp parallel for schedule(static)          /* rows: what cv::parallel_for_ does in pyrDown */
    for (int y = 0; y < D->h; y++) {
        for (int x = 0; x < D->w; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) {
                const int sy = reflect101(2 * y + j - 2, sh);
                int row = 0;
                for (int i = 0; i < 5; i++) {
                    const int sx = reflect101(2 * x + i - 2, sw);
                    row += k[i] * (int)*gray_at(S, sx, sy);
                }
                acc += k[j] * row;
            }
            *gray_at(D, x, y) = (uint8_t)((acc + 128) >> 8);
        }
    }
}

/* lkpyramid.cpp ScharrDerivInvoker: vertical [3 10 3] / [-1 0 1], then horizontal
 * [-1 0 1] / [3 10 3]; rows and columns mirrored REFLECT_101 on the ROI. */
static void scharr_deriv(orc_level *L)
{
    const int w = L->w, h = L->h;
#pragma omp parallel                                   /* ScharrDerivInvoker is a parallel_for_ over rows */
    {
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
    int *t1 = (int *)malloc(sizeof(int) * (size_t)(w + 2));
#pragma omp for schedule(static)
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = gray_at(L, 0, y > 0 ? y - 1 : (h > 1 ? 1 : 0));
        const uint8_t *r1 = gray_at(L, 0, y);
        const uint8_t *r2 = gray_at(L, 0, y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0));
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        const int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        int16_t *d = deriv_at(L, 0, y);
        for (int x = 0; x < w; x++) {
            d[2 * x]     = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
    }
}

/* cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), maxLevel, withDerivatives=true,
 *   pyrBorder=BORDER_REFLECT_101, derivBorder=BORDER_CONSTANT)
 * (reference call site: src/tracker/image_pyramid.cpp:42-46). Returns the pyramid;
 * pyr->nlevels - 1 is the value OpenCV returns. */
orc_pyramid *orc_pyramid_build(const uint8_t *img, int w, int h, int stride, int win, int max_level)
{
    if (!img || w <= 0 || h <= 0 || win <= 2 || max_level < 0 || max_level >= ORC_MAX_LEVELS) return NULL;
    orc_pyramid *P = (orc_pyramid *)calloc(1, sizeof(orc_pyramid));
    if (!P) return NULL;
    int lw = w, lh = h;
    for (int level = 0; level <= max_level; level++) {
        orc_level *L = &P->lv[level];
        if (level_alloc(L, lw, lh, win)) return NULL;
        if (level == 0) {
            for (int y = 0; y < lh; y++) memcpy(gray_at(L, 0, y), img + (size_t)y * stride, (size_t)lw);
        } else {
            pyr_down(&P->lv[level - 1], L);
        }
        gray_make_border(L);
        scharr_deriv(L);            /* deriv border stays 0 (calloc) */
        P->nlevels = level + 1;
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        if (lw <= win || lh <= win) break;
    }
    return P;
}

void orc_pyramid_free(orc_pyramid *P)
{
    if (!P) return;
    for (int i = 0; i < ORC_MAX_LEVELS; i++) { free(P->lv[i].gray); free(P->lv[i].deriv); }
    free(P);
}

int orc_pyramid_levels(const orc_pyramid *P) { return P ? P->nlevels : 0; }

int orc_pyramid_level_size(const orc_pyramid *P, int level, int *w, int *h)
{
    if (!P || level < 0 || level >= P->nlevels) return -1;
    *w = P->lv[level].w; *h = P->lv[level].h;
    return 0;
}

/* Copy a level out. padded != 0: full (w+2*win) x (h+2*win) OpenCV layout, else interior. */
int orc_pyramid_copy_gray(const orc_pyramid *P, int level, int padded, uint8_t *out)
{
    if (!P || level < 0 || level >= P->nlevels) return -1;
    const orc_level *L = &P->lv[level];
    const int p = padded ? L->pad : 0;
    const int ow = L->w + 2 * p;
    for (int y = -p; y < L->h + p; y++)
        memcpy(out + (size_t)(y + p) * ow, gray_at(L, -p, y), (size_t)ow);
    return 0;
}

int orc_pyramid_copy_deriv(const orc_pyramid *P, int level, int padded, int16_t *out)
{
    if (!P || level < 0 || level >= P->nlevels) return -1;
    const orc_level *L = &P->lv[level];
    const int p = padded ? L->pad : 0;
    const int ow = 2 * (L->w + 2 * p);
    for (int y = -p; y < L->h + p; y++)
        memcpy(out + (size_t)(y + p) * ow, deriv_at(L, -p, y), sizeof(int16_t) * (size_t)ow);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Lucas-Kanade                                                               */
/* ------------------------------------------------------------------------- */

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
#define ORC_USE_INITIAL_FLOW 4      /* cv::OPTFLOW_USE_INITIAL_FLOW */

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* round-half-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

static inline void bilinear_weights(float a, float b, int *iw00, int *iw01, int *iw10, int *iw11)
{
    *iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    *iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
    *iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
    *iw11 = (1 << W_BITS) - *iw00 - *iw01 - *iw10;
}


enum { ORC_ACC_INT64 = 0, ORC_ACC_F32_SCALAR = 1, ORC_ACC_F32_SIMD128 = 2, ORC_ACC_F32_SIMD128_FMA = 3,
       ORC_ACC_F32_SSE2_LEGACY = 4, ORC_ACC_F32_WIDE8 = 5, ORC_ACC_MODES = 6 };

/* A11/A12/A22 of one window in the float order of `mode` (lkpyramid.cpp LKTrackerInvoker, "extract the patch from the
 * first image, compute covariation matrix of derivatives"). dI = interleaved [Ix, Iy] int16, row-major win x win. */
static void acc_A_f32(int mode, const int16_t *dI, int win, float *A11, float *A12, float *A22)
{
    float s11 = 0.f, s12 = 0.f, s22 = 0.f;                   /* `acctype iA11` = float: the scalar (tail) accumulator */
    const int lanes = mode == ORC_ACC_F32_WIDE8 ? 8 : 4;
    const int step = mode == ORC_ACC_F32_SSE2_LEGACY ? 4 : 8;   /* pixels consumed per SIMD loop trip */
    float q11[8] = {0}, q12[8] = {0}, q22[8] = {0};
    const int fused = mode == ORC_ACC_F32_SIMD128_FMA;
    for (int y = 0; y < win; y++) {
        int x = 0;
        if (mode != ORC_ACC_F32_SCALAR)
            for (; x <= win - step; x += step)
                for (int g = 0; g < step; g += lanes)          /* SIMD128: two 4-lane groups per trip, first pixels x..x+3 */
                    for (int l = 0; l < lanes; l++) {
                        const float fx = (float)dI[2 * (y * win + x + g + l)], fy = (float)dI[2 * (y * win + x + g + l) + 1];
                        if (fused) { q22[l] = fmaf(fy, fy, q22[l]); q12[l] = fmaf(fx, fy, q12[l]); q11[l] = fmaf(fx, fx, q11[l]); }
                        else { q22[l] = q22[l] + fy * fy; q12[l] = q12[l] + fx * fy; q11[l] = q11[l] + fx * fx; }
                    }
        for (; x < win; x++) {                                  /* iA11 += (itemtype)(ixval*ixval) */
            const int ix = dI[2 * (y * win + x)], iy = dI[2 * (y * win + x) + 1];
            s11 += (float)(ix * ix); s12 += (float)(ix * iy); s22 += (float)(iy * iy);
        }
    }
    if (mode == ORC_ACC_F32_SIMD128 || mode == ORC_ACC_F32_SIMD128_FMA) {        /* iA11 += v_reduce_sum(qA11) */
        s11 += (q11[0] + q11[2]) + (q11[1] + q11[3]); s12 += (q12[0] + q12[2]) + (q12[1] + q12[3]);
        s22 += (q22[0] + q22[2]) + (q22[1] + q22[3]);
    } else if (mode == ORC_ACC_F32_SSE2_LEGACY) {                                /* iA11 += buf[0] + buf[1] + buf[2] + buf[3] */
        s11 += q11[0] + q11[1] + q11[2] + q11[3]; s12 += q12[0] + q12[1] + q12[2] + q12[3]; s22 += q22[0] + q22[1] + q22[2] + q22[3];
    } else if (mode == ORC_ACC_F32_WIDE8) {                                      /* halves folded, then the 4-lane reduce */
        float h11[4], h12[4], h22[4];
        for (int l = 0; l < 4; l++) { h11[l] = q11[l] + q11[l + 4]; h12[l] = q12[l] + q12[l + 4]; h22[l] = q22[l] + q22[l + 4]; }
        s11 += (h11[0] + h11[2]) + (h11[1] + h11[3]); s12 += (h12[0] + h12[2]) + (h12[1] + h12[3]); s22 += (h22[0] + h22[2]) + (h22[1] + h22[3]);
    }
    *A11 = s11; *A12 = s12; *A22 = s22;
}

/* b1/b2 of one iteration: diff = It (int, |diff| <= 8160) per pixel, row-major win x win, against dI. In every SIMD variant
 * v_dotprod / _mm_madd_epi16 forms It_i*I_i + It_(i+4)*I_(i+4) exactly in int32 (i = 0..3 of an 8-pixel step), converts that to
 * f32 and adds it into lane pair (qb0: i = 0, 1; qb1: i = 2, 3); the end is (qb0 + qb1) lane-wise, then lanes 0 + 2 (x) and
 * 1 + 3 (y) added to the scalar tail accumulator. */
static void acc_b_f32(int mode, const int *diff, const int16_t *dI, int win, float *b1, float *b2)
{
    float sb1 = 0.f, sb2 = 0.f;
    float qx[8] = {0}, qy[8] = {0};                             /* [0..1] = qb0's pixel pairs, [2..3] = qb1's (WIDE8: 8 pairs) */
    const int half = mode == ORC_ACC_F32_WIDE8 ? 8 : 4, step = 2 * half;
    for (int y = 0; y < win; y++) {
        int x = 0;
        if (mode != ORC_ACC_F32_SCALAR)
            for (; x <= win - step; x += step)
                for (int i = 0; i < half; i++) {
                    const int p0 = y * win + x + i, p1 = p0 + half;
                    qx[i] = qx[i] + (float)(diff[p0] * dI[2 * p0] + diff[p1] * dI[2 * p1]);
                    qy[i] = qy[i] + (float)(diff[p0] * dI[2 * p0 + 1] + diff[p1] * dI[2 * p1 + 1]);
                }
        for (; x < win; x++) {
            const int p = y * win + x;
            sb1 += (float)(diff[p] * dI[2 * p]); sb2 += (float)(diff[p] * dI[2 * p + 1]);
        }
    }
    if (mode == ORC_ACC_F32_WIDE8) {
        for (int i = 0; i < 4; i++) { qx[i] = qx[i] + qx[i + 4]; qy[i] = qy[i] + qy[i + 4]; }
    }
    if (mode != ORC_ACC_F32_SCALAR) {
        /* qb0 + qb1: lanes [x(0)+x(2), y(0)+y(2), x(1)+x(3), y(1)+y(3)]; reduce: lane0 + lane2 / lane1 + lane3 */
        sb1 += (qx[0] + qx[2]) + (qx[1] + qx[3]);
        sb2 += (qy[0] + qy[2]) + (qy[1] + qy[3]);
    }
    *b1 = sb1; *b2 = sb2;
}

/* test hooks: the two accumulations on a caller-supplied window (tests/test_oracle_lk_accumulators.py pins them to an
 * independent numpy binary32 evaluation of the same lane orders) */
void orc_lk_acc_A(int mode, const int16_t *dI, int win, float *out3) { acc_A_f32(mode, dI, win, out3, out3 + 1, out3 + 2); }
void orc_lk_acc_b(int mode, const int *diff, const int16_t *dI, int win, float *out2) { acc_b_f32(mode, diff, dI, win, out2, out2 + 1); }

/* One pyramid level of LKTrackerInvoker::operator() for all points.
 * iters_out (optional): Gauss-Newton iterations executed per point at this level. */
static void lk_level(const orc_level *I, const orc_level *J, int npts,
                     const float *prev_pts, float *next_pts, uint8_t *status, float *err,
                     int win, int level, int max_level, int max_count, double epsilon,
                     int flags, float min_eig_threshold, int *iters_out, int acc_mode)
{
    const float half_win = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
#pragma omp parallel                                   /* LKTrackerInvoker is a parallel_for_ over points */
    {
    int16_t *Iwin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win);
    int16_t *dIwin = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win * 2);
    int *diffwin = (int *)malloc(sizeof(int) * (size_t)win * win);

#pragma omp for schedule(dynamic, 4)
    for (int pt = 0; pt < npts; pt++) {
        if (iters_out) iters_out[pt] = 0;
        const float lscale = (float)(1. / (double)(1 << level));
        float prev_x = prev_pts[2 * pt] * lscale, prev_y = prev_pts[2 * pt + 1] * lscale;
        float next_x, next_y;
        if (level == max_level) {
            if (flags & ORC_USE_INITIAL_FLOW) {
                next_x = next_pts[2 * pt] * lscale; next_y = next_pts[2 * pt + 1] * lscale;
            } else {
                next_x = prev_x; next_y = prev_y;
            }
        } else {
            next_x = next_pts[2 * pt] * 2.f; next_y = next_pts[2 * pt + 1] * 2.f;
        }
        next_pts[2 * pt] = next_x; next_pts[2 * pt + 1] = next_y;

        prev_x -= half_win; prev_y -= half_win;
        const int ipx = cv_floor_f(prev_x), ipy = cv_floor_f(prev_y);

        if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
            if (level == 0) { status[pt] = 0; err[pt] = 0; }
            continue;
        }

        float a = prev_x - (float)ipx, b = prev_y - (float)ipy;
        int iw00, iw01, iw10, iw11;
        bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);

        int64_t iA11 = 0, iA12 = 0, iA22 = 0;
        for (int y = 0; y < win; y++) {
            const uint8_t *src0 = gray_at(I, ipx, ipy + y), *src1 = gray_at(I, ipx, ipy + y + 1);
            const int16_t *d0 = deriv_at(I, ipx, ipy + y), *d1 = deriv_at(I, ipx, ipy + y + 1);
            for (int x = 0; x < win; x++) {
                const int ival = DESCALE(src0[x] * iw00 + src0[x + 1] * iw01 +
                                         src1[x] * iw10 + src1[x + 1] * iw11, W_BITS - 5);
                const int ixval = DESCALE(d0[2 * x] * iw00 + d0[2 * x + 2] * iw01 +
                                          d1[2 * x] * iw10 + d1[2 * x + 2] * iw11, W_BITS);
                const int iyval = DESCALE(d0[2 * x + 1] * iw00 + d0[2 * x + 3] * iw01 +
                                          d1[2 * x + 1] * iw10 + d1[2 * x + 3] * iw11, W_BITS);
                Iwin[y * win + x] = (int16_t)ival;
                dIwin[2 * (y * win + x)] = (int16_t)ixval;
                dIwin[2 * (y * win + x) + 1] = (int16_t)iyval;
                iA11 += (int64_t)(ixval * ixval);
                iA12 += (int64_t)(ixval * iyval);
                iA22 += (int64_t)(iyval * iyval);
            }
        }

        float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        if (acc_mode != ORC_ACC_INT64) {
            acc_A_f32(acc_mode, dIwin, win, &A11, &A12, &A22);
            A11 *= FLT_SCALE; A12 *= FLT_SCALE; A22 *= FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                              (float)(2 * win * win);

        if (min_eig < min_eig_threshold || D < FLT_EPSILON) {
            if (level == 0) status[pt] = 0;
            continue;
        }
        D = 1.f / D;

        next_x -= half_win; next_y -= half_win;
        float prev_dx = 0.f, prev_dy = 0.f;

        for (int j = 0; j < max_count; j++) {
            const int inx = cv_floor_f(next_x), iny = cv_floor_f(next_y);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                if (level == 0) status[pt] = 0;
                break;
            }
            if (iters_out) iters_out[pt] = j + 1;

            a = next_x - (float)inx; b = next_y - (float)iny;
            bilinear_weights(a, b, &iw00, &iw01, &iw10, &iw11);

            int64_t ib1 = 0, ib2 = 0;
            for (int y = 0; y < win; y++) {
                const uint8_t *J0 = gray_at(J, inx, iny + y), *J1 = gray_at(J, inx, iny + y + 1);
                for (int x = 0; x < win; x++) {
                    const int diff = DESCALE(J0[x] * iw00 + J0[x + 1] * iw01 +
                                             J1[x] * iw10 + J1[x + 1] * iw11, W_BITS - 5) - Iwin[y * win + x];
                    ib1 += (int64_t)(diff * dIwin[2 * (y * win + x)]);
                    ib2 += (int64_t)(diff * dIwin[2 * (y * win + x) + 1]);
                    diffwin[y * win + x] = diff;
                }
            }
            float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
            if (acc_mode != ORC_ACC_INT64) {
                acc_b_f32(acc_mode, diffwin, dIwin, win, &b1, &b2);
                b1 *= FLT_SCALE; b2 *= FLT_SCALE;
            }
            const float dx = (float)((A12 * b2 - A22 * b1) * D);
            const float dy = (float)((A12 * b1 - A11 * b2) * D);

            next_x += dx; next_y += dy;
            next_pts[2 * pt] = next_x + half_win; next_pts[2 * pt + 1] = next_y + half_win;

            if ((double)dx * (double)dx + (double)dy * (double)dy <= epsilon) break;

            if (j > 0 && fabs((double)(dx + prev_dx)) < 0.01 && fabs((double)(dy + prev_dy)) < 0.01) {
                next_pts[2 * pt] -= dx * 0.5f; next_pts[2 * pt + 1] -= dy * 0.5f;
                break;
            }
            prev_dx = dx; prev_dy = dy;
        }

        /* err != NULL and OPTFLOW_LK_GET_MIN_EIGENVALS unset (optical_flow.cpp:46-49):
         * the level-0 epilogue recomputes the window and can clear status. */
        if (status[pt] && level == 0) {
            const float nx = next_pts[2 * pt] - half_win, ny = next_pts[2 * pt + 1] - half_win;
            const int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
            if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                status[pt] = 0;
                continue;
            }
            const float aa = nx - (float)inx, bb = ny - (float)iny;
            bilinear_weights(aa, bb, &iw00, &iw01, &iw10, &iw11);
            float errval = 0.f;
            for (int y = 0; y < win; y++) {
                const uint8_t *J0 = gray_at(J, inx, iny + y), *J1 = gray_at(J, inx, iny + y + 1);
                for (int x = 0; x < win; x++) {
                    const int diff = DESCALE(J0[x] * iw00 + J0[x + 1] * iw01 +
                                             J1[x] * iw10 + J1[x + 1] * iw11, W_BITS - 5) - Iwin[y * win + x];
                    errval += fabsf((float)diff);
                }
            }
            err[pt] = errval * 1.f / (float)(32 * win * win);
        }
    }
    free(Iwin); free(dIwin); free(diffwin);
    }
}

/* threads used by the row / point loops above (1 = the scalar port; 0 = all host cores) */
#ifdef _OPENMP
#include <omp.h>
int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); else omp_set_num_threads(omp_get_num_procs()); return omp_get_max_threads(); }
#else
int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prevPts, nextPts, status, err, Size(win,win),
 *   maxLevel, TermCriteria(COUNT|EPS, maxCount, eps), flags, minEigThreshold)
 * (reference call site: src/tracker/optical_flow.cpp:46-49).
 * next_pts is in/out (input only when flags & ORC_USE_INITIAL_FLOW).
 * iters_out (optional): [nlevels][npts] iteration counts (level-major). */
int orc_klt_track_mode(const orc_pyramid *prev, const orc_pyramid *next, int npts,
                       const float *prev_pts, float *next_pts, uint8_t *status, float *err,
                       int win, int max_level, int max_count, double eps, int flags,
                       double min_eig_threshold, int *iters_out, int acc_mode);

int orc_klt_track(const orc_pyramid *prev, const orc_pyramid *next, int npts,
                  const float *prev_pts, float *next_pts, uint8_t *status, float *err,
                  int win, int max_level, int max_count, double eps, int flags,
                  double min_eig_threshold, int *iters_out)
{
    return orc_klt_track_mode(prev, next, npts, prev_pts, next_pts, status, err, win, max_level, max_count, eps, flags,
                              min_eig_threshold, iters_out, ORC_ACC_INT64);
}

/* acc_mode: ORC_ACC_* (see the header). ORC_ACC_INT64 is the oracle the HIP kernel is checked against. */
int orc_klt_track_mode(const orc_pyramid *prev, const orc_pyramid *next, int npts,
                       const float *prev_pts, float *next_pts, uint8_t *status, float *err,
                       int win, int max_level, int max_count, double eps, int flags,
                       double min_eig_threshold, int *iters_out, int acc_mode)
{
    if (!prev || !next || acc_mode < 0 || acc_mode >= ORC_ACC_MODES) return -1;
    if (max_level > prev->nlevels - 1) max_level = prev->nlevels - 1;
    if (max_level > next->nlevels - 1) max_level = next->nlevels - 1;
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (eps < 0.) eps = 0.;
    if (eps > 10.) eps = 10.;
    const double epsilon = eps * eps;
    for (int i = 0; i < npts; i++) { status[i] = 1; err[i] = 0.f; }
    for (int level = max_level; level >= 0; level--) {
        lk_level(&prev->lv[level], &next->lv[level], npts, prev_pts, next_pts, status, err,
                 win, level, max_level, max_count, epsilon, flags, (float)min_eig_threshold,
                 iters_out ? iters_out + (size_t)level * npts : NULL, acc_mode);
    }
    return 0;
}

/* tracker::Feature::Status (src/tracker/track.hpp:9-21) */
enum { ST_TRACKED = 0, ST_NEW = 1, ST_FAILED_FLOW = 2, ST_RANSAC_OUTLIER = 3, ST_FLOW_OUT_OF_RANGE = 4 };

/* computeImplementationCpu (src/tracker/optical_flow.cpp:10-59): runs LK, maps
 * status 0 -> FAILED_FLOW else TRACKED, then overrides to FLOW_OUT_OF_RANGE if the
 * corner left the level-0 image. corners is in/out; track_status is int32 per point. */
int orc_optical_flow_compute(const orc_pyramid *prev, const orc_pyramid *cur, int npts,
                             const float *prev_corners, float *corners, int32_t *track_status,
                             int use_initial_corners, int win, int max_level, int max_iter,
                             double eps, double min_eig_threshold)
{
    if (npts == 0) return 0;
    uint8_t *st = (uint8_t *)malloc((size_t)npts);
    float *err = (float *)malloc(sizeof(float) * (size_t)npts);
    const int rc = orc_klt_track(prev, cur, npts, prev_corners, corners, st, err, win, max_level,
                                 max_iter, eps, use_initial_corners ? ORC_USE_INITIAL_FLOW : 0,
                                 min_eig_threshold, NULL);
    const int width = cur->lv[0].w, height = cur->lv[0].h;
    for (int i = 0; i < npts && rc == 0; i++) {
        const float x = corners[2 * i], y = corners[2 * i + 1];
        track_status[i] = st[i] == 0 ? ST_FAILED_FLOW : ST_TRACKED;
        if (x < 0.0f || x >= (float)width || y < 0.0f || y >= (float)height)
            track_status[i] = ST_FLOW_OUT_OF_RANGE;
    }
    free(st); free(err);
    return rc;
}
