// GFTT feature detector: min-eigenvalue corner response + per-block arg-max.
//
// Replaces the detector HybVIO runs on the CPU (featureDetector "GPU-GFTT" without a GPU image
// factory): CpuCornerResponse = cv::cornerMinEigenVal(gray, response, gfttBlockSize, 3)
// (src/tracker/feature_detector.cpp:279-315), CollectMax::cpuImplementation (:393-417) and the
// sort / zero-prefix / applyMinDistance tail of detect() (:610-634, feature_detector_legacy.cpp:
// 177-213). Algorithm and float evaluation order: oracle/gftt_oracle.c.
//
// Design for CDNA4: the reference materialises five full-size float images (Dx, Dy, three products)
// plus the response before it scans for block maxima. Here one workgroup owns one bs x bs block of
// one image: it stages the (bs+4)^2 gray neighbourhood in LDS (BORDER_REFLECT_101), forms the
// derivative products and their 3x3 box sums in two register-tiled passes, reduces 16*response to the block's arg-max
// (first maximum in raster order, as the reference's scan) and writes ONE 12-byte key point. The
// level-0 image already sits in HBM for the pyramid: the detector's HBM traffic is that image read
// once (+27 % halo); the response map never exists. Sorting the <= (w/bs)(h/bs) key points and the
// greedy min-distance filter are the reference's host code and stay on the host.
#include <algorithm>
#include <vector>

#include "hv_internal.hpp"

namespace hv {

namespace {

struct GfttArgs {
    const uint8_t *const *l0_ptr;     // per-slot level-0 image table (filled by the pyramid build)
    const int *l0_stride;
    const int *slots;                 // [n_images], or null: the single image slot0
    int slot0;
    int w, h, nbx, nby;
    float k0, k1, min_response;
    float *kp;                        // [n_images][nby * nbx][3] = x, y, response
    int n_images;
};

// total order of the reference's raster scan with strict '>': higher response wins, ties -> lower index
__device__ __forceinline__ bool better(float ra, int ia, float rb, int ib) { return ra > rb || (ra == rb && ia < ib); }

typedef float float4v __attribute__((ext_vector_type(4)));

// Two register-tiled passes over LDS instead of one LDS round trip per stencil stage:
//   pass 1  task (tile row ty in [-1, BS], strip of 4 columns): 3 rows x 8 gray values -> the products of
//           6 columns -> the 3 row sums of 4 columns (12 floats, three 16-byte LDS stores)
//   pass 2  task (row y, strip of 4 columns): 3 rows x 3 channels of row sums (nine 16-byte LDS loads) ->
//           4 responses -> running arg-max
// ~45 VALU + 5 LDS instructions per pixel. Positions outside the image take the products of their
// BORDER_REFLECT_101 mirror (that is what box-filtering the product images does): rows by evaluating
// the task at the mirrored centre row, columns by copying the mirrored column inside the strip.
template <int BS>
__global__ __launch_bounds__(256) void gftt_block_kernel(GfttArgs a)
{
    constexpr int GW = BS + 4;                       // gray tile edge: tile column c <-> image x0 - 2 + c
    constexpr int NS = BS / 4;                       // 4-column strips per row
    __shared__ __attribute__((aligned(16))) float gray[GW * GW];
    __shared__ __attribute__((aligned(16))) float rsum[3][(BS + 2) * BS];      // row sums, tile rows -1 .. BS
    __shared__ float red_r[4];
    __shared__ int red_i[4];

    const int t = threadIdx.x;
    const int blocks = a.nbx * a.nby;
    // neighbouring blocks share halo rows and 128-byte lines: keep them on one XCD's L2 (rocprof: 2.7x the
    // image bytes were fetched from HBM with the default round-robin placement)
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int img = lb / blocks, bi = lb - img * blocks;
    const int yb = bi / a.nbx, xb = bi - yb * a.nbx;
    const int x0 = xb * BS, y0 = yb * BS, w = a.w, h = a.h;
    const int slot = a.slots ? a.slots[img] : a.slot0;
    const uint8_t *src = a.l0_ptr[slot];
    const int stride = a.l0_stride[slot];

    // ---- stage the gray tile as floats: one (unaligned) 8-byte load and two 16-byte LDS stores per task where
    // the 8 pixels are inside the image, per-byte BORDER_REFLECT_101 otherwise ----
    {
        constexpr int NC = (GW + 7) / 8;              // 8-pixel chunks per tile row (the last one may be half used)
        for (int i = t; i < GW * NC; i += 256) {
            const int ty = i / NC, q = i - ty * NC;
            const uint8_t *row = src + (size_t)reflect101(y0 - 2 + ty, h) * stride;
            const int x = x0 - 2 + 8 * q;             // image column of byte 0 of this chunk
            uint32_t lo, hi;
            if (x >= 0 && x + 8 <= w) {
                uint2 v;
                __builtin_memcpy(&v, row + x, 8);
                lo = v.x; hi = v.y;
            } else {
                lo = hi = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lo |= (uint32_t)row[reflect101(x + k, w)] << (8 * k);
                    hi |= (uint32_t)row[reflect101(x + 4 + k, w)] << (8 * k);
                }
            }
            float *dst = &gray[ty * GW + 8 * q];
            *reinterpret_cast<float4v *>(dst) = float4v{(float)(lo & 0xFFu), (float)((lo >> 8) & 0xFFu), (float)((lo >> 16) & 0xFFu), (float)(lo >> 24)};
            if (8 * q + 4 < GW)
                *reinterpret_cast<float4v *>(dst + 4) = float4v{(float)(hi & 0xFFu), (float)((hi >> 8) & 0xFFu), (float)((hi >> 16) & 0xFFu), (float)(hi >> 24)};
        }
    }
    __syncthreads();

    // ---- pass 1: row sums of the derivative products ----
    const float k0 = a.k0, k1 = a.k1;
    for (int task = t; task < (BS + 2) * NS; task += 256) {
        const int ry = task / NS, sx = task - ry * NS;                 // tile row ry - 1, columns 4 sx .. 4 sx + 3
        const int cy = reflect101(y0 + ry - 1, h) - y0 + 2;           // gray tile row of the (mirrored) centre row
        // gray tile columns 4 sx .. 4 sx + 7  <->  block columns 4 sx - 2 .. 4 sx + 5
        float g[3][8];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float4v lo = *reinterpret_cast<const float4v *>(&gray[(cy - 1 + j) * GW + 4 * sx]);
            const float4v hi = *reinterpret_cast<const float4v *>(&gray[(cy - 1 + j) * GW + 4 * sx + 4]);
#pragma unroll
            for (int k = 0; k < 4; k++) { g[j][k] = lo[k]; g[j][4 + k] = hi[k]; }
        }
        float c0[6], c1[6], c2[6];                                      // products at block columns 4 sx - 1 .. 4 sx + 4
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const float dt = g[0][k + 2] - g[0][k], dm = g[1][k + 2] - g[1][k], db = g[2][k + 2] - g[2][k];
            const float vx = k0 * dm + k1 * (dt + db);
            const float st = k0 * g[0][k + 1] + k1 * (g[0][k] + g[0][k + 2]);
            const float sb = k0 * g[2][k + 1] + k1 * (g[2][k] + g[2][k + 2]);
            const float vy = sb - st;
            c0[k] = vx * vx; c1[k] = vx * vy; c2[k] = vy * vy;
        }
        // columns outside the image: block column -1 mirrors +1 (first strip of a block at x = 0), column BS
        // mirrors BS - 2 (last strip of a block that ends at the image edge)
        if (sx == 0 && x0 == 0) { c0[0] = c0[2]; c1[0] = c1[2]; c2[0] = c2[2]; }
        if (sx == NS - 1 && x0 + BS == w) { c0[5] = c0[3]; c1[5] = c1[3]; c2[5] = c2[3]; }
        float4v r0, r1, r2;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r0[k] = (c0[k] + c0[k + 1]) + c0[k + 2];
            r1[k] = (c1[k] + c1[k + 1]) + c1[k + 2];
            r2[k] = (c2[k] + c2[k + 1]) + c2[k + 2];
        }
        *reinterpret_cast<float4v *>(&rsum[0][ry * BS + 4 * sx]) = r0;
        *reinterpret_cast<float4v *>(&rsum[1][ry * BS + 4 * sx]) = r1;
        *reinterpret_cast<float4v *>(&rsum[2][ry * BS + 4 * sx]) = r2;
    }
    __syncthreads();

    // ---- pass 2: column sums, min eigenvalue, arg-max in raster order ----
    float best_r = -1e10f;
    int best_i = 0;
    for (int task = t; task < BS * NS; task += 256) {
        const int y = task / NS, sx = task - y * NS;
        float4v s[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float4v up = *reinterpret_cast<const float4v *>(&rsum[ch][y * BS + 4 * sx]);          // tile row y - 1
            const float4v mid = *reinterpret_cast<const float4v *>(&rsum[ch][(y + 1) * BS + 4 * sx]);
            const float4v dn = *reinterpret_cast<const float4v *>(&rsum[ch][(y + 2) * BS + 4 * sx]);
            s[ch] = (up + mid) + dn;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float aa = s[0][k] * 0.5f, bb = s[1][k], cc = s[2][k] * 0.5f, amc = aa - cc;
            const float resp = (aa + cc) - sqrtf(amc * amc + bb * bb);
            const float r16 = resp * 16.0f;                          // CpuCornerResponse::GAIN
            if (r16 > best_r && r16 > a.min_response) { best_r = r16; best_i = y * BS + 4 * sx + k; }
        }
    }
    // threads without a candidate keep (-1e10, 0): index 0 / response -1e10 is also the reference's "no corner"
    for (int o = 32; o > 0; o >>= 1) {
        const float ro = __shfl_down(best_r, o);
        const int io = __shfl_down(best_i, o);
        if (better(ro, io, best_r, best_i)) { best_r = ro; best_i = io; }
    }
    if ((t & 63) == 0) { red_r[t >> 6] = best_r; red_i[t >> 6] = best_i; }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 4; k++) if (better(red_r[k], red_i[k], best_r, best_i)) { best_r = red_r[k]; best_i = red_i[k]; }
        const bool found = best_r > -1e10f;
        float *o = a.kp + ((size_t)img * blocks + bi) * 3;
        o[0] = found ? (float)(x0 + best_i % BS) : 0.f;
        o[1] = found ? (float)(y0 + best_i / BS) : 0.f;
        o[2] = best_r;
    }
}


// ---- gfttBlockSize 5 / 7 (feature_detector.cpp:279-315 passes it to cv::cornerMinEigenVal as the box size; the default is 3 and
// the two kernels above are built for it). A plain three-pass kernel, one workgroup per arg-max block: product images of the
// (BS + 2 hb)^2 neighbourhood, each product evaluated at its BORDER_REFLECT_101 mirrored centre (= box-filtering the reflected
// product images, as cv::boxFilter does); row sums left to right; column sums top to bottom; response; arg-max. The oracle's float
// order (oracle/gftt_oracle.c). Not tuned: non-default configurations only.
template <int BS>
__global__ __launch_bounds__(256) void gftt_box_kernel(GfttArgs a, int hb)
{
    constexpr int HBMAX = 3, GWMAX = BS + 2 * (HBMAX + 1), NPMAX = BS + 2 * HBMAX;
    __shared__ float gray[GWMAX * GWMAX];
    __shared__ float prod[3][NPMAX * NPMAX];
    __shared__ float rsum[3][NPMAX * BS];
    __shared__ float red_r[4];
    __shared__ int red_i[4];
    const int t = threadIdx.x;
    const int blocks = a.nbx * a.nby;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int img = lb / blocks, bi = lb - img * blocks;
    const int yb = bi / a.nbx, xb = bi - yb * a.nbx;
    const int x0 = xb * BS, y0 = yb * BS, w = a.w, h = a.h;
    const int slot = a.slots ? a.slots[img] : a.slot0;
    const uint8_t *src = a.l0_ptr[slot];
    const int stride = a.l0_stride[slot];
    const int G = hb + 1, GW = BS + 2 * G, NP = BS + 2 * hb;
    for (int i = t; i < GW * GW; i += 256) {
        const int r = i / GW, c = i - r * GW;
        gray[i] = (float)src[(size_t)reflect101(y0 - G + r, h) * stride + reflect101(x0 - G + c, w)];
    }
    __syncthreads();
    const float k0 = a.k0, k1 = a.k1;
    for (int i = t; i < NP * NP; i += 256) {
        const int pr = i / NP, pc = i - pr * NP;
        // centre of product position (y0 - hb + pr, x0 - hb + pc), mirrored into the image, in tile coordinates
        const int cy = reflect101(y0 - hb + pr, h) - y0 + G, cx = reflect101(x0 - hb + pc, w) - x0 + G;
        const float *g = gray + cy * GW + cx;
        const float dt = g[-GW + 1] - g[-GW - 1], dm = g[1] - g[-1], db = g[GW + 1] - g[GW - 1];
        const float vx = k0 * dm + k1 * (dt + db);
        const float st = k0 * g[-GW] + k1 * (g[-GW - 1] + g[-GW + 1]);
        const float sb = k0 * g[GW] + k1 * (g[GW - 1] + g[GW + 1]);
        const float vy = sb - st;
        prod[0][i] = vx * vx; prod[1][i] = vx * vy; prod[2][i] = vy * vy;
    }
    __syncthreads();
    for (int i = t; i < NP * BS; i += 256) {
        const int pr = i / BS, x = i - pr * BS;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float *p = prod[ch] + pr * NP + x;          // product columns x - hb .. x + hb  <->  indices x .. x + 2 hb
            float s_ = p[0];
            for (int k = 1; k <= 2 * hb; k++) s_ = s_ + p[k];
            rsum[ch][i] = s_;
        }
    }
    __syncthreads();
    float best_r = -1e10f;
    int best_i = 0;
    for (int i = t; i < BS * BS; i += 256) {
        const int y = i / BS, x = i - y * BS;
        float sm[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float *p = rsum[ch] + y * BS + x;           // product rows y - hb .. y + hb  <->  rsum rows y .. y + 2 hb
            float s_ = p[0];
            for (int k = 1; k <= 2 * hb; k++) s_ = s_ + p[k * BS];
            sm[ch] = s_;
        }
        const float aa = sm[0] * 0.5f, bb = sm[1], cc = sm[2] * 0.5f, amc = aa - cc;
        const float resp = (aa + cc) - sqrtf(amc * amc + bb * bb);
        const float r16 = resp * 16.0f;                          // CpuCornerResponse::GAIN
        if (r16 > best_r && r16 > a.min_response) { best_r = r16; best_i = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ro = __shfl_down(best_r, o);
        const int io = __shfl_down(best_i, o);
        if (better(ro, io, best_r, best_i)) { best_r = ro; best_i = io; }
    }
    if ((t & 63) == 0) { red_r[t >> 6] = best_r; red_i[t >> 6] = best_i; }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 4; k++) if (better(red_r[k], red_i[k], best_r, best_i)) { best_r = red_r[k]; best_i = red_i[k]; }
        const bool found = best_r > -1e10f;
        float *o = a.kp + ((size_t)img * blocks + bi) * 3;
        o[0] = found ? (float)(x0 + best_i % BS) : 0.f;
        o[1] = found ? (float)(y0 + best_i / BS) : 0.f;
        o[2] = best_r;
    }
}

#ifndef GFTT_MARCH_UNROLL
#define GFTT_MARCH_UNROLL 2   // 6 removes the window moves but needs 147 VGPRs (3 waves per SIMD): 0.93 ms against 0.84
#endif

// ---- r02: the same arithmetic without LDS or barriers. A thread owns a strip of 4 columns of one arg-max block and marches
// down its BS + 2 product rows: the horizontal Sobel pieces (difference d, smoothed value s) of a gray row are formed once
// and slide through a 3-row register window, the row sums of the three product images slide through a second one, so every
// gray row is loaded once per strip (8 bytes, next row requested one step ahead) and every intermediate is computed once
// (the tiled kernel above re-forms the horizontal pieces three times and moves 72 bytes of LDS per pixel: 105 VALU
// instructions per pixel measured, and it is VALU-issue bound). The NS = BS / 4 strips of a block are consecutive lanes: the
// block arg-max is a 3-step lane exchange. Blocks on the first / last image rows evaluate each product row at its mirrored
// centre with three fresh gray rows (no sliding); columns outside the image mirror inside the strip, as above.
template <int BS>
__global__ __launch_bounds__(256) void gftt_march_kernel(GfttArgs a)
{
    constexpr int NS = BS / 4;
    const int blocks = a.nbx * a.nby;
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    const long long gid = (long long)wg * 256 + threadIdx.x;
    const long long total = (long long)a.n_images * blocks * NS;
    const bool live = gid < total;
    const long long gidc = live ? gid : total - 1;
    const int sx = (int)(gidc % NS);
    const long long bg = gidc / NS;
    const int img = (int)(bg / blocks), bi = (int)(bg - (long long)img * blocks);
    const int yb = bi / a.nbx, xb = bi - yb * a.nbx;
    const int x0 = xb * BS, y0 = yb * BS, w = a.w, h = a.h;
    const int slot = a.slots ? a.slots[img] : a.slot0;
    const uint8_t *src = a.l0_ptr[slot];
    const int stride = a.l0_stride[slot];
    const float k0 = a.k0, k1 = a.k1;

    const int xs = x0 - 2 + 4 * sx;                                   // image column of byte 0 of this strip's 8-byte row segment
    const bool col_in = xs >= 0 && xs + 8 <= w;
    int cxr[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cxr[k] = reflect101(xs + k, w);
    auto load_row = [&](int gr) -> uint2 {                            // gray row gr (inside the image), columns xs .. xs + 7
        const uint8_t *row = src + (size_t)gr * stride;
        uint2 v;
        if (col_in) {
            __builtin_memcpy(&v, row + xs, 8);
        } else {
            v.x = v.y = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v.x |= (uint32_t)row[cxr[k]] << (8 * k);
                v.y |= (uint32_t)row[cxr[4 + k]] << (8 * k);
            }
        }
        return v;
    };
    struct HRow { float d[6], s[6]; };
    auto hrow = [&](uint2 v) -> HRow {
        float g[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { g[k] = (float)((v.x >> (8 * k)) & 0xFFu); g[4 + k] = (float)((v.y >> (8 * k)) & 0xFFu); }
        HRow o;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            o.d[k] = g[k + 2] - g[k];
            o.s[k] = k0 * g[k + 1] + k1 * (g[k] + g[k + 2]);
        }
        return o;
    };

    // rows y0 - 2 .. y0 + BS + 1 all exist: consecutive centres, the window slides
    const bool slide = y0 >= 2 && y0 + BS + 1 <= h - 1;
    const bool mirror_l = sx == 0 && x0 == 0, mirror_r = sx == NS - 1 && x0 + BS == w;
    HRow up, mid, dn;
    float4v rsA[3], rsB[3];                                          // row sums of product rows ry - 2, ry - 1
    float best_r = -1e10f;
    int best_i = 0;
    uint2 nxt = make_uint2(0, 0);
    if (slide) {
        up = hrow(load_row(y0 - 2));
        mid = hrow(load_row(y0 - 1));
        nxt = load_row(y0);
    }
#pragma unroll GFTT_MARCH_UNROLL
    for (int ry = -1; ry <= BS; ++ry) {
        if (slide) {
            if (ry > -1) { up = mid; mid = dn; }
            dn = hrow(nxt);
            if (ry < BS) nxt = load_row(y0 + ry + 2);                 // the row the next step appends
        } else {
            const int cy = reflect101(y0 + ry, h);
            const uint2 ru = load_row(reflect101(cy - 1, h)), rm = load_row(cy), rd = load_row(reflect101(cy + 1, h));
            up = hrow(ru); mid = hrow(rm); dn = hrow(rd);
        }
        float c0[6], c1[6], c2[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const float vx = k0 * mid.d[k] + k1 * (up.d[k] + dn.d[k]);
            const float vy = dn.s[k] - up.s[k];
            c0[k] = vx * vx; c1[k] = vx * vy; c2[k] = vy * vy;
        }
        if (mirror_l) { c0[0] = c0[2]; c1[0] = c1[2]; c2[0] = c2[2]; }
        if (mirror_r) { c0[5] = c0[3]; c1[5] = c1[3]; c2[5] = c2[3]; }
        float4v r[3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r[0][k] = (c0[k] + c0[k + 1]) + c0[k + 2];
            r[1][k] = (c1[k] + c1[k + 1]) + c1[k + 2];
            r[2][k] = (c2[k] + c2[k + 1]) + c2[k + 2];
        }
        if (ry >= 1) {
            const int y = ry - 1;
            float4v sm[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) sm[ch] = (rsA[ch] + rsB[ch]) + r[ch];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float aa = sm[0][k] * 0.5f, bb = sm[1][k], cc = sm[2][k] * 0.5f, amc = aa - cc;
                const float resp = (aa + cc) - sqrtf(amc * amc + bb * bb);
                const float r16 = resp * 16.0f;                          // CpuCornerResponse::GAIN
                if (r16 > best_r && r16 > a.min_response) { best_r = r16; best_i = y * BS + 4 * sx + k; }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { rsA[ch] = rsB[ch]; rsB[ch] = r[ch]; }
    }
    // the NS strips of a block are NS consecutive lanes (NS divides 64): butterfly over them
#pragma unroll
    for (int o = 1; o < NS; o <<= 1) {
        const float ro = __shfl_xor(best_r, o);
        const int io = __shfl_xor(best_i, o);
        if (better(ro, io, best_r, best_i)) { best_r = ro; best_i = io; }
    }
    if (live && sx == 0) {
        const bool found = best_r > -1e10f;
        float *o = a.kp + ((size_t)img * blocks + bi) * 3;
        o[0] = found ? (float)(x0 + best_i % BS) : 0.f;
        o[1] = found ? (float)(y0 + best_i / BS) : 0.f;
        o[2] = best_r;
    }
}

int launch(Ctx *c, int n_images, const int *slots_dev, int slot0, int bs, float min_response, int block_size, float *kp_dev)
{
    if (block_size != 3 && block_size != 5 && block_size != 7) return HV_ERR_UNSUPPORTED;   // odd box sizes up to 7 (default 3)
    GfttArgs a{};
    a.l0_ptr = c->d_l0_ptr; a.l0_stride = c->d_l0_stride; a.slots = slots_dev; a.slot0 = slot0;
    a.w = c->L.w[0]; a.h = c->L.h[0];
    a.nbx = a.w / bs; a.nby = a.h / bs;                          // std::ceil(int / int): the ragged edge is skipped
    const double scale = 1.0 / (4.0 * block_size * 255.0);
    a.k1 = (float)scale; a.k0 = (float)(2.0 * scale);
    a.min_response = min_response; a.kp = kp_dev;
    const unsigned grid = (unsigned)(a.nbx * a.nby * n_images);
    if (grid == 0) return HV_OK;
    a.n_images = n_images;
    ScopedKernelTime tm(c, HV_K_GFTT);
    if (block_size != 3) {                                       // non-default box size: the plain kernel
        if (bs == 32)      hipLaunchKernelGGL(gftt_box_kernel<32>, dim3(grid), dim3(256), 0, c->stream, a, block_size / 2);
        else if (bs == 16) hipLaunchKernelGGL(gftt_box_kernel<16>, dim3(grid), dim3(256), 0, c->stream, a, block_size / 2);
        else if (bs == 8)  hipLaunchKernelGGL(gftt_box_kernel<8>, dim3(grid), dim3(256), 0, c->stream, a, block_size / 2);
        else return HV_ERR_INVALID;
        HV_HIP(c, hipGetLastError());
        return HV_OK;
    }
    // The marching kernel is the throughput design (a thread walks 34 dependent steps); for a handful of images the LDS-tiled
    // kernel (one workgroup per block, 256 threads side by side) finishes sooner -- 25 us per frame in the bench latency leg.
    // knob gftt_tiled (tests / experiments only): 1 / 0 forces the tiled / the marching kernel.
    const int force_tiled = c->knob.gftt_tiled;
    const bool tiled = force_tiled >= 0 ? force_tiled != 0 : n_images < 128;
    if (!tiled && a.w >= 8 && a.h >= 3) {
        const long long threads = (long long)grid * (bs / 4);
        const unsigned wgs = (unsigned)((threads + 255) / 256);
        if (bs == 32)      hipLaunchKernelGGL(gftt_march_kernel<32>, dim3(wgs), dim3(256), 0, c->stream, a);
        else if (bs == 16) hipLaunchKernelGGL(gftt_march_kernel<16>, dim3(wgs), dim3(256), 0, c->stream, a);
        else if (bs == 8)  hipLaunchKernelGGL(gftt_march_kernel<8>, dim3(wgs), dim3(256), 0, c->stream, a);
        else return HV_ERR_INVALID;
        HV_HIP(c, hipGetLastError());
        return HV_OK;
    }
    if (bs == 32)      hipLaunchKernelGGL(gftt_block_kernel<32>, dim3(grid), dim3(256), 0, c->stream, a);
    else if (bs == 16) hipLaunchKernelGGL(gftt_block_kernel<16>, dim3(grid), dim3(256), 0, c->stream, a);
    else if (bs == 8)  hipLaunchKernelGGL(gftt_block_kernel<8>, dim3(grid), dim3(256), 0, c->stream, a);
    else return HV_ERR_INVALID;
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace

}  // namespace hv

using hv::Ctx;

extern "C" {

void hv_gftt_default_params(hv_gftt_params *p)
{
    if (!p) return;
    p->gfttBlockSize = 3; p->gfttMinDistance = 50.0; p->gfttMinResponse = 1e-3f; p->maxTracks = 200;
}

int hv_gftt_block_size(const hv_gftt_params *p)
{
    if (!p) return HV_ERR_INVALID;
    const int target = (int)p->gfttMinDistance;                  // feature_detector.cpp:428-436
    return target >= 32 ? 32 : target >= 16 ? 16 : 8;
}

int hv_gftt_keypoint_count(hv_ctx *ctx, const hv_gftt_params *p)
{
    if (!ctx || !p) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    const int bs = hv_gftt_block_size(p);
    return (c->L.w[0] / bs) * (c->L.h[0] / bs);
}

int hv_gftt_keypoints_batch_dev(hv_ctx *ctx, const hv_gftt_params *p, int n_images, const int *slots_dev, float *kp_dev)
{
    if (!ctx || !p || n_images < 0 || (n_images > 0 && (!slots_dev || !kp_dev))) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    return hv::launch(c, n_images, slots_dev, 0, hv_gftt_block_size(p), p->gfttMinResponse, p->gfttBlockSize, kp_dev);
}

void hv_apply_min_distance(float *corners_xy, int *n_inout, const float *prev_xy, int n_prev, int r, int max_tracks)
{
    // FeatureDetector::applyMinDistance (feature_detector_legacy.cpp:177-213): in place, in order
    if (!corners_xy || !n_inout) return;
    int n_out = 0;
    const float r2 = static_cast<float>(r * r);
    for (int k = 0; k < *n_inout; ++k) {
        const float cx = corners_xy[2 * k], cy = corners_xy[2 * k + 1];
        bool near_other = false;
        if (r > 0) {
            for (int i = 0; i < n_prev && !near_other; ++i) {
                const float dx = prev_xy[2 * i] - cx, dy = prev_xy[2 * i + 1] - cy;
                near_other = dx * dx + dy * dy < r2;
            }
            for (int i = 0; i < n_out && !near_other; ++i) {
                const float dx = corners_xy[2 * i] - cx, dy = corners_xy[2 * i + 1] - cy;
                near_other = dx * dx + dy * dy < r2;
            }
        }
        if (!near_other) { corners_xy[2 * n_out] = cx; corners_xy[2 * n_out + 1] = cy; ++n_out; }
        if (n_out >= max_tracks) break;
    }
    *n_inout = n_out;
}

int hv_gftt_detect(hv_ctx *ctx, const hv_gftt_params *p, int slot, const float *prev_xy, int n_prev, int mask_radius,
                   float *corners_xy, int capacity, int *n_out)
{
    if (!ctx || !p || !corners_xy || !n_out || n_prev < 0 || (n_prev > 0 && !prev_xy)) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    if (slot < 0 || slot >= c->p.pool_size || !c->slot_used[slot]) return HV_ERR_POOL;
    const int nk = hv_gftt_keypoint_count(ctx, p);
    if (capacity < 2 * nk) return HV_ERR_INVALID;
    *n_out = 0;
    if (nk == 0) return HV_OK;
    if (c->gftt_cap < nk) {
        if (c->d_gftt_kp) { (void)hipFree(c->d_gftt_kp); c->d_gftt_kp = nullptr; c->gftt_cap = 0; }
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_gftt_kp), sizeof(float) * 3 * nk));
        c->gftt_cap = nk;
    }
    int rc = hv::launch(c, 1, nullptr, slot, hv_gftt_block_size(p), p->gfttMinResponse, p->gfttBlockSize, c->d_gftt_kp);
    if (rc != HV_OK) return rc;
    std::vector<float> kp(3 * (size_t)nk);
    HV_HIP(c, hipMemcpyAsync(kp.data(), c->d_gftt_kp, sizeof(float) * 3 * nk, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    // detect(): stable sort by descending response, then corners.resize(n) + push_back (n zero points first)
    std::vector<int> order(nk);
    for (int i = 0; i < nk; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kp[3 * a + 2] > kp[3 * b + 2]; });
    for (int i = 0; i < nk; ++i) { corners_xy[2 * i] = 0.f; corners_xy[2 * i + 1] = 0.f; }
    for (int i = 0; i < nk; ++i) { corners_xy[2 * (nk + i)] = kp[3 * order[i]]; corners_xy[2 * (nk + i) + 1] = kp[3 * order[i] + 1]; }
    int n = 2 * nk;
    if (mask_radius > 0) hv_apply_min_distance(corners_xy, &n, prev_xy, n_prev, mask_radius, p->maxTracks);
    *n_out = n;
    return HV_OK;
}

}  // extern "C"
